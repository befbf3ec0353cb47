// raster_binning.hip -- tile binning of the gfx950 rasterizer: offsets, instance duplication,
// stable LSD radix sort of (tile | depth) keys, per-tile ranges.
//
//   scan_block_sums_kernel  second half of the InclusiveSum (rasterizer_impl.cu:278); the first
//                           half (per-workgroup totals) is fused into the preprocess kernel and
//                           the intra-workgroup scan into the duplicate kernel, so the scan costs
//                           one tiny launch instead of a device-wide pass.
//   duplicate_kernel        duplicateWithKeys (rasterizer_impl.cu:70-111)
//   radix_*                 replaces cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:304-309).
//                           Hand-written for wave64: per-wave digit matching with 64-bit ballots,
//                           9-bit digits over the key's LIVE bits only (31 depth bits + the tile id's
//                           bits: 41..44 -> 5 passes; cub walks all 43..46 bits in 6).  Measured per pass at
//                           1.9 M keys: 51 / 59 / 69 / 111 us for 8 / 9 / 10 / 11-bit digits.  The result is the unique stable
//                           order, i.e. bit-identical to any other stable sort on the same bits.
//   tile_ranges_kernel      identifyTileRanges (rasterizer_impl.cu:116-138)
//
// All of this is integer / byte work bounded by HBM and launch latency, not by ALU.
#include "raster_common.h"

namespace gd {

uint32_t higher_msb(uint32_t n)  // getHigherMsb, rasterizer_impl.cu:35-50
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

SortPlan plan_sort(uint32_t tiles_total)
{
    SortPlan p;
    p.total_bits = 32 + (int)higher_msb(tiles_total);      // what the reference hands to cub (rasterizer_impl.cu:304-309)
    // Bits that can differ between two keys: the 31 low bits of the depth (depths are > 0.2, in_frustum, so the sign
    // bit of every key is clear) and the bits of the largest tile id.  A stable LSD sort on just those gives the same
    // permutation as one on total_bits -- 8 views x 512^2: 44 live bits = 5 passes of 9 instead of 46 = 5 of 10.
    int tile_bits = 0;
    while (tile_bits < 31 && (1u << tile_bits) < tiles_total) tile_bits++;
    p.live_bits = 31 + tile_bits;
    p.passes = (p.live_bits + kMaxDigitBits - 1) / kMaxDigitBits;
    p.digit_bits = (p.live_bits + p.passes - 1) / p.passes;
    if (p.digit_bits < 8) p.digit_bits = 8;
    return p;
}

namespace {

// Exclusive scan of block_sums[0..n) in place; block_sums[n] receives the grand total (= R).
// Sync-free form (info != NULL, round 5): the total stays on the device -- info[0] = R, info[1] = the instance count the
// binning kernels work on (R if it fits the caller's capacity, else 0: nothing is rendered), info[2] = 1 on overflow.
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(uint32_t* __restrict__ sums, uint32_t n,
                                                               uint32_t* __restrict__ info, uint32_t capacity)
{
    __shared__ uint32_t wave_incl[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? sums[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += t;
        }
        if (lane == 63) wave_incl[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (uint32_t w = 0; w < wave; w++) wave_off += wave_incl[w];
        const uint32_t carry = carry_s;
        if (i < n) sums[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        sums[n] = carry_s;
        if (info) {
            const bool over = carry_s > capacity;
            info[0] = carry_s;
            info[1] = over ? 0u : carry_s;
            info[2] = over ? 1u : 0u;
            info[3] = capacity;
        }
    }
}

__global__ __launch_bounds__(kGaussBlock) void duplicate_kernel(int VP, int P, const int* __restrict__ radii,
                                                                GeomState gs, uint64_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out,
                                                                uint32_t* __restrict__ slot_vp,
                                                                uint4* __restrict__ rowpos, uint32_t gx, uint32_t gy,
                                                                const uint32_t* __restrict__ info)
{
    __shared__ uint32_t wave_incl[kGaussBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vp = blockIdx.x * kGaussBlock + tid;
    const uint32_t touched = vp < VP ? gs.tiles_touched[vp] : 0;
    uint32_t incl = touched;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += t;
    }
    if (lane == 63) wave_incl[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (uint32_t w = 0; w < wave; w++) wave_off += wave_incl[w];
    const uint32_t incl_global = gs.block_sums[blockIdx.x] + wave_off + incl;
    if (vp >= VP) return;
    gs.point_offsets[vp] = incl_global;  // inclusive, as the reference stores it
    if (info && info[2]) {               // sync-free form: more instances than the binning buffer holds -- write none,
        gs.tiles_touched[vp] = 0;        // and leave the backward pass (instance_sum_kernel) nothing to gather either
        return;
    }
    const int r = radii[vp];
    if (r > 0) {
        uint32_t off = incl_global - touched;
        const float2 xy = gs.means2D[vp];
        uint32_t x0, y0, x1, y1;
        tile_rect(xy.x, xy.y, r, gx, gy, x0, y0, x1, y1);
        const uint32_t depth_bits = __float_as_uint(gs.rgbd[vp].w);
        const uint32_t view_tile0 = (uint32_t)(vp / P) * gx * gy;
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) {
                uint64_t key = (uint64_t)(view_tile0 + y * gx + x);
                key <<= 32;
                key |= depth_bits;
                keys_out[off] = key;
                // the sort carries the instance's SLOT (its index here), not the Gaussian: the backward blend stores its
                // per-instance rows BY SLOT, so that a Gaussian's rows are contiguous and can be added without atomics;
                // the Gaussian of a slot is kept in slot_vp (tile_ranges_kernel turns point_list into Gaussian ids)
                vals_out[off] = off;
                if (slot_vp) slot_vp[off] = (uint32_t)vp;
                if (rowpos) rowpos[off] = make_uint4(0, 0, 0, 0);   // "no strip blended this instance" until render_forward says otherwise
                off++;
            }
    }
}

// ---------------- radix sort ----------------
// Element i of a workgroup tile belongs to wave i / (64*kSortItems); inside the wave the order
// is (step, lane).  Stability follows from ranking in exactly that order.

// digit of the key with the always-zero depth sign bit squeezed out: live bit i is key bit i (i < 31) or i + 1
__device__ __forceinline__ uint32_t live_digit(uint64_t key, int shift, uint32_t mask)
{
    const uint64_t live = (key & 0x7fffffffull) | ((key >> 32) << 31);
    return (uint32_t)(live >> shift) & mask;
}

template <int BITS>
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, int shift,
                                                         uint32_t mask, uint32_t* __restrict__ hist, uint32_t nblk,
                                                         const uint32_t* __restrict__ n_dev)
{
    constexpr int BINS = 1 << BITS;
    if (n_dev) n = *n_dev;      // sync-free form: the key count lives on the device, the grid covers the capacity
    __shared__ uint32_t h[BINS];
    for (int i = threadIdx.x; i < BINS; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t idx = base + k * 256 + threadIdx.x;
        if (idx < n) atomicAdd(&h[live_digit(keys[idx], shift, mask)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS; i += 256) hist[(size_t)i * nblk + blockIdx.x] = h[i];
}

// One workgroup per digit: exclusive scan of hist[d][0..nblk) in place, digit total -> totals[d].
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ hist, uint32_t nblk,
                                                         uint32_t* __restrict__ totals)
{
    __shared__ uint32_t wave_incl[4];
    __shared__ uint32_t carry_s;
    uint32_t* row = hist + (size_t)blockIdx.x * nblk;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblk; base += 256) {
        const uint32_t i = base + tid;
        const uint32_t v = i < nblk ? row[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += t;
        }
        if (lane == 63) wave_incl[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (uint32_t w = 0; w < wave; w++) wave_off += wave_incl[w];
        const uint32_t carry = carry_s;
        if (i < nblk) row[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (tid == 255) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry_s;
}

template <int BITS>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint64_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                            uint32_t mask, const uint32_t* __restrict__ hist,
                                                            const uint32_t* __restrict__ totals, uint32_t nblk,
                                                            const uint32_t* __restrict__ n_dev)
{
    constexpr int BINS = 1 << BITS;
    constexpr int PER_THREAD = BINS / 256;
    if (n_dev) {
        n = *n_dev;
        if (blockIdx.x * kSortTile >= n) return;     // (workgroup-uniform) nothing of this tile is live
    }
    __shared__ uint32_t cnt[4][BINS];    // per-wave running digit counts, later wave bases
    __shared__ uint32_t dbase[BINS];     // global base of digit d for this workgroup
    __shared__ uint32_t wave_tot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // exclusive scan of digit totals (BINS entries) + this workgroup's offset inside each digit
    uint32_t loc[PER_THREAD];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {
        loc[k] = totals[tid * PER_THREAD + k];
        sum += loc[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    for (int i = tid; i < 4 * BINS; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < wave; w++) run += wave_tot[w];
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {
        const int d = tid * PER_THREAD + k;
        dbase[d] = run + hist[(size_t)d * nblk + blockIdx.x];
        run += loc[k];
    }
    __syncthreads();

    // rank every key inside its wave
    const uint32_t base = blockIdx.x * kSortTile + wave * (64 * kSortItems);
    uint64_t key[kSortItems];
    uint32_t rank[kSortItems];
    volatile uint32_t* wc = cnt[wave];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < kSortItems; s++) {
        const uint32_t idx = base + s * 64 + lane;
        const bool valid = idx < n;
        key[s] = valid ? keys_in[idx] : ~0ull;
        const uint32_t d = live_digit(key[s], shift, mask);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        uint32_t r = 0;
        if (valid) {
            const uint32_t c = wc[d];
            r = c + before;
            if (before == 0) wc[d] = c + (uint32_t)__popcll(peers);
        }
        rank[s] = r;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // turn per-wave counts into per-wave bases (exclusive over waves) + global digit base
    for (int d = tid; d < BINS; d += 256) {
        uint32_t run2 = dbase[d];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t c = cnt[w][d];
            cnt[w][d] = run2;
            run2 += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSortItems; s++) {
        const uint32_t idx = base + s * 64 + lane;
        if (idx < n) {
            const uint32_t d = live_digit(key[s], shift, mask);
            const uint32_t pos = cnt[wave][d] + rank[s];
            keys_out[pos] = key[s];
            vals_out[pos] = vals_in[idx];
        }
    }
}

__global__ void tile_ranges_kernel(const uint64_t* __restrict__ keys, uint32_t L, uint2* __restrict__ ranges,
                                   uint32_t* __restrict__ point_list, const uint32_t* __restrict__ slot_vp,
                                   uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ n_dev)
{
    if (n_dev) L = *n_dev;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    if (slot_vp) {   // the sort's epilogue: the payload (slot) moves aside, point_list becomes the Gaussian ids
        const uint32_t slot = point_list[idx];
        slot_of[idx] = slot;
        point_list[idx] = slot_vp[slot];
    }
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) {
            ranges[prev].y = idx;
            ranges[cur].x = idx;
        }
    }
    if (idx == L - 1) ranges[cur].y = L;
}

// perm = the launch order of the blend kernels' workgroups: longest tile lists first, XCD by XCD.  Workgroup b runs on XCD
// b % 8 (observed; a speed hint only) and every XCD has its own L2, so -- as in the index order it replaces -- XCD x keeps the
// contiguous BAND x of n / 8 tiles (neighbouring tiles share most of their Gaussians); inside a band the tiles go by descending
// list length (buckets of 8 entries; the order inside a bucket is whatever the atomics give: it only decides WHEN a tile's
// workgroup runs, never a result).  perm[8 r + x] = the r-th longest tile of band x.  n % 8 != 0: one band, plain descending order.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint2* __restrict__ ranges, uint32_t n, uint32_t* __restrict__ perm)
{
    __shared__ uint32_t cnt[8][256], off[8][256];
    const uint32_t tid = threadIdx.x;
    const uint32_t nb = (n & 7u) == 0u ? 8u : 1u, per = n / nb;
    for (uint32_t i = tid; i < 8u * 256u; i += 1024u) (&cnt[0][0])[i] = 0u;
    __syncthreads();
    for (uint32_t t = tid; t < n; t += 1024u) {
        const uint2 r = ranges[t];
        atomicAdd(&cnt[t / per][255u - min((r.y - r.x) >> 3, 255u)], 1u);
    }
    __syncthreads();
    for (uint32_t i = tid; i < nb * 256u; i += 1024u) {     // exclusive scan per band: serial over the predecessors (tiny)
        const uint32_t band = i >> 8, bk = i & 255u;
        uint32_t acc = 0u;
        for (uint32_t j = 0; j < bk; j++) acc += cnt[band][j];
        off[band][bk] = acc;
    }
    __syncthreads();
    for (uint32_t t = tid; t < n; t += 1024u) {
        const uint2 r = ranges[t];
        const uint32_t band = t / per;
        const uint32_t rank = atomicAdd(&off[band][255u - min((r.y - r.x) >> 3, 255u)], 1u);
        perm[rank * nb + band] = t;
    }
}

// ---------------- tile-bucketed binning (round 6) ----------------
// The reference sorts all R (tile | depth) keys globally (cub radix sort, rasterizer_impl.cu:304-309); rounds 1-5 did the same
// with an own 5-pass LSD radix sort: ~17 launches, 0.35 ms per 8-view launch, the largest rasterizer item.  But the tile half
// of the key is known before any sorting: count the instances of every tile, prefix-sum the counts (= the reference's
// `ranges`, identifyTileRanges :116-138, for free), scatter each instance into its tile's bucket, and sort every bucket on its
// own in LDS.  The result is the unique order (tile, depth, Gaussian index) -- bit-identical `keys`, `point_list`, `ranges`.

// counts[tile] (kept in ranges[tile].x, zeroed by the launcher) += 1 per instance.  Device-scope atomics are performed at
// the memory side on this multi-die part (every XCD has its own L2) and sixteen counters share a cache line: one global
// atomic per instance measured 0.30 ms per 8-view launch (3.5 M atomics).  So a workgroup of 1024 consecutive (view, Gaussian)
// pairs -- at most two views -- first counts into an LDS window of its two views' tiles, then adds the non-zero cells: ~7x
// fewer global atomics.  Images with more tiles per view than the window holds take the direct form.
constexpr uint32_t kAggThreads = 1024;
constexpr uint32_t kAggWindow = 8192;        // LDS counters: 2 views x up to 4096 tiles (1024 x 1024 pixels)

__device__ __forceinline__ bool agg_rect(int vp, int VP, int P, const int* __restrict__ radii, const GeomState& gs, uint32_t gx,
                                         uint32_t gy, uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1, uint32_t& view)
{
    if (vp >= VP) return false;
    const int r = radii[vp];
    if (r <= 0) return false;
    const float2 xy = gs.means2D[vp];
    tile_rect(xy.x, xy.y, r, gx, gy, x0, y0, x1, y1);
    view = (uint32_t)(vp / P);
    return true;
}

template <bool LDS_WINDOW>
__global__ __launch_bounds__(kAggThreads) void tile_count_kernel(int VP, int P, const int* __restrict__ radii, GeomState gs,
                                                                 uint2* __restrict__ ranges, uint32_t gx, uint32_t gy)
{
    __shared__ uint32_t h[LDS_WINDOW ? kAggWindow : 1];
    const uint32_t tpv = gx * gy;
    const int vp = blockIdx.x * kAggThreads + threadIdx.x;
    const uint32_t view0 = (uint32_t)((blockIdx.x * kAggThreads) / (uint32_t)P);
    if (LDS_WINDOW) {
        for (uint32_t i = threadIdx.x; i < 2 * tpv; i += kAggThreads) h[i] = 0;
        __syncthreads();
    }
    uint32_t x0, y0, x1, y1, view;
    if (agg_rect(vp, VP, P, radii, gs, gx, gy, x0, y0, x1, y1, view)) {
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) {
                if (LDS_WINDOW) atomicAdd(&h[(view - view0) * tpv + y * gx + x], 1u);
                else atomicAdd(&ranges[view * tpv + y * gx + x].x, 1u);
            }
    }
    if (LDS_WINDOW) {
        __syncthreads();
        const uint32_t tiles_total_hi = (uint32_t)((VP + P - 1) / P) * tpv;
        for (uint32_t i = threadIdx.x; i < 2 * tpv; i += kAggThreads) {
            const uint32_t c = h[i], t = view0 * tpv + i;
            if (c && t < tiles_total_hi) atomicAdd(&ranges[t].x, c);
        }
    }
}

// The scatter of the tile-bucketed binning with the same aggregation: count into the LDS window, reserve a contiguous piece of
// every touched tile's bucket with ONE global atomic per (workgroup, tile), then hand out the piece's positions with LDS atomics.
// Also does duplicate_kernel's bookkeeping: point_offsets (the inclusive scan, from the 256-Gaussian block sums), slot_vp.
template <bool LDS_WINDOW>
__global__ __launch_bounds__(kAggThreads) void tile_scatter_kernel(int VP, int P, const int* __restrict__ radii, GeomState gs,
                                                                   uint64_t* __restrict__ bkeys, uint32_t* __restrict__ slot_vp,
                                                                   uint32_t* __restrict__ cursor, uint32_t gx, uint32_t gy,
                                                                   const uint32_t* __restrict__ info)
{
    __shared__ uint32_t h[LDS_WINDOW ? kAggWindow : 1];
    __shared__ uint32_t wave_incl[kAggThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tpv = gx * gy;
    const int vp = blockIdx.x * kAggThreads + tid;
    const uint32_t view0 = (uint32_t)((blockIdx.x * kAggThreads) / (uint32_t)P);
    const uint32_t touched = vp < VP ? gs.tiles_touched[vp] : 0;
    uint32_t incl = touched;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += t;
    }
    if (lane == 63) wave_incl[wave] = incl;
    if (LDS_WINDOW)
        for (uint32_t i = tid; i < 2 * tpv; i += kAggThreads) h[i] = 0;
    __syncthreads();
    // block_sums are per kGaussBlock = 256 Gaussians = 4 waves: this workgroup covers four of those blocks
    uint32_t wave_off = 0;
    for (uint32_t w = wave & ~3u; w < wave; w++) wave_off += wave_incl[w];
    const uint32_t blk = (uint32_t)vp / kGaussBlock;
    const uint32_t incl_global = vp < VP ? gs.block_sums[blk] + wave_off + incl : 0;
    const bool voided = info && info[2];       // sync-free form: capacity or bucket overflow -- nothing is binned
    if (vp < VP) {
        gs.point_offsets[vp] = incl_global;    // inclusive, as the reference stores it
        if (voided) gs.tiles_touched[vp] = 0;  // ... and the backward pass (instance_sum_kernel) gathers nothing either
    }
    uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0, view = 0;
    const bool live = !voided && agg_rect(vp, VP, P, radii, gs, gx, gy, x0, y0, x1, y1, view);
    if (LDS_WINDOW) {
        if (live)
            for (uint32_t y = y0; y < y1; y++)
                for (uint32_t x = x0; x < x1; x++) atomicAdd(&h[(view - view0) * tpv + y * gx + x], 1u);
        __syncthreads();
        const uint32_t tiles_total_hi = (uint32_t)((VP + P - 1) / P) * tpv;
        for (uint32_t i = tid; i < 2 * tpv; i += kAggThreads) {
            const uint32_t c = h[i], t = view0 * tpv + i;
            if (c && t < tiles_total_hi) h[i] = atomicAdd(&cursor[t], c);      // this workgroup's piece of the bucket starts here
        }
        __syncthreads();
    }
    if (!live) return;
    uint32_t off = incl_global - touched;
    const uint32_t depth_bits = __float_as_uint(gs.rgbd[vp].w);
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0; x < x1; x++) {
            const uint32_t pos = LDS_WINDOW ? atomicAdd(&h[(view - view0) * tpv + y * gx + x], 1u)
                                            : atomicAdd(&cursor[view * tpv + y * gx + x], 1u);
            // key = depth bits << 32 | slot: the per-tile sort orders by depth and, for equal depths, by slot = by Gaussian
            // index, which is the order the reference's stable sort leaves (duplicateWithKeys emits in index order)
            bkeys[pos] = ((uint64_t)depth_bits << 32) | off;
            slot_vp[off] = (uint32_t)vp;
            off++;
        }
}

// One workgroup: exclusive scan of the tile counts -> ranges[t] = [start, end) ((0, 0) for an empty tile, as the reference's
// zero-initialised array keeps it), cursor[t] = start, stats = {total, longest list}.  Sync-free form (info != NULL): a
// capacity overflow (info[2] == 1, scan_block_sums_kernel) or a list beyond kBucketMax (-> info[2] = 2, nothing is binned:
// every view shows the background, the caller repeats the call on the radix path) leaves every range empty.
__global__ __launch_bounds__(1024) void tile_scan_kernel(uint2* __restrict__ ranges, uint32_t n, uint32_t* __restrict__ cursor,
                                                         uint32_t* __restrict__ stats, uint32_t* __restrict__ info)
{
    __shared__ uint32_t wave_incl[16], wave_max[16];
    __shared__ uint32_t carry_s, void_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // longest list first (decides whether anything is binned at all)
    uint32_t mx = 0;
    for (uint32_t i = tid; i < n; i += 1024) mx = max(mx, ranges[i].x);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
    if (lane == 0) wave_max[wave] = mx;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        for (int w = 0; w < 16; w++) m = max(m, wave_max[w]);
        uint32_t vd = 0;
        if (info) {
            if (info[2]) vd = 1;
            else if (m > kBucketMax) { info[2] = 2u; info[1] = 0u; vd = 1; }
        }
        stats[1] = m;
        void_s = vd;
    }
    __syncthreads();
    const bool nothing = void_s != 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = (i < n && !nothing) ? ranges[i].x : 0;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += t;
        }
        if (lane == 63) wave_incl[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (uint32_t w = 0; w < wave; w++) wave_off += wave_incl[w];
        const uint32_t carry = carry_s;
        if (i < n) {
            const uint32_t start = carry + wave_off + incl - v;
            ranges[i] = v ? make_uint2(start, start + v) : make_uint2(0u, 0u);
            cursor[i] = start;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (tid == 0) stats[0] = carry_s;
}

// One workgroup per tile sorts the bucket's keys (depth bits << 32 | slot; unique, so the order is unique) and writes what the
// radix path's sort + tile_ranges_kernel leave: keys[pos] = tile << 32 | depth bits, slot_of[pos] = slot, point_list[pos] = the
// slot's Gaussian.
//
// Not a comparison network: the first two forms of this kernel were bitonic networks in LDS (one level per barrier: 0.21 ms per
// 8-view launch; three levels per LDS round trip with compile-time strides: 0.195 ms) and both were VALU-bound -- O(n log^2 n)
// compare-exchanges are ~90 M wave instructions per launch on the benchmark scene, as many as the backward blend issues.
// This form is a COUNTING sort on a monotone quantisation of the depth, made exact by ranking inside the quantisation cells:
//   1. min / max of the tile's depth bits (positive floats order like their bit patterns);
//   2. cell q = floor((d - min) * B / (max - min + 1)) in [0, B), B = the list length rounded up to a power of two (>= 256), so a
//      cell holds one key on average; the map is monotone (integer -> float conversion, multiplication by a positive constant
//      and truncation all are), hence keys of different cells are already in their final relative order;
//   3. count per cell (LDS atomics, low half of a 32-bit word), exclusive scan (high half = the cell's start), scatter into the
//      cell (LDS atomic on the low half again: position = start + old count) -- the order INSIDE a cell is whatever the atomics give;
//   4. every key ranks itself among the m keys of its cell by full 64-bit comparison (m is 1-2 typically; two thin depth
//      clusters at the ends of the range -- a garment's front and back -- give m of a few dozen; all depths equal gives m = n:
//      n^2 / 256 comparisons per thread, ~15 us for that one tile at n = 1700: slow, never wrong, never serial) and writes its
//      outputs at start + rank.
// ~60 instructions per key instead of ~1300; 48 KiB of LDS (keys 32 KiB, cells 16 KiB): three workgroups per CU.
__global__ __launch_bounds__(256) void tile_sort_kernel(const uint2* __restrict__ ranges, const uint64_t* __restrict__ bkeys,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list,
                                                        uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ slot_vp)
{
    __shared__ uint64_t sk[kBucketMax];        // the keys, grouped by cell
    __shared__ uint32_t cell[kBucketMax];      // per cell: count (low 16 bits) | start (high 16 bits)
    __shared__ uint32_t red[8];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n == 0) return;
    uint32_t B = 256;
    while (B < n) B <<= 1;
    // the thread's keys (i = tid + 256 e) stay in registers: ONE trip to global memory instead of one per pass
    constexpr int kPer = kBucketMax / 256;
    uint64_t mine[kPer];
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        const uint32_t i = tid + 256u * e;
        mine[e] = i < n ? bkeys[rg.x + i] : ~0ull;
    }
    // 1. depth range
    uint32_t dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        if (tid + 256u * e < n) {
            const uint32_t d = (uint32_t)(mine[e] >> 32);
            dmin = min(dmin, d);
            dmax = max(dmax, d);
        }
    }
    for (uint32_t i = tid; i < B; i += 256) cell[i] = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, off, 64));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off, 64));
    }
    if (lane == 0) { red[wave] = dmin; red[4 + wave] = dmax; }
    __syncthreads();
    dmin = min(min(red[0], red[1]), min(red[2], red[3]));
    dmax = max(max(red[4], red[5]), max(red[6], red[7]));
    const float scale = (float)B / (float)(dmax - dmin + 1u);
    auto cell_of = [&](uint32_t d) { return min(B - 1u, (uint32_t)((float)(d - dmin) * scale)); };
    // 2. + 3a. counts
#pragma unroll
    for (int e = 0; e < kPer; e++)
        if (tid + 256u * e < n) atomicAdd(&cell[cell_of((uint32_t)(mine[e] >> 32))], 1u);
    __syncthreads();
    // 3b. exclusive scan over the B cells: thread t owns the epb consecutive cells t * epb ...
    {
        const uint32_t epb = B >> 8;
        uint32_t local = 0;
        for (uint32_t e = 0; e < epb; e++) local += cell[tid * epb + e];
        uint32_t incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += t;
        }
        __syncthreads();                       // red[] is read above by every thread
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        uint32_t run = incl - local;
        for (uint32_t w = 0; w < wave; w++) run += red[w];
        for (uint32_t e = 0; e < epb; e++) {
            const uint32_t c = cell[tid * epb + e];
            cell[tid * epb + e] = run << 16;   // count restarts at 0: the scatter below counts again
            run += c;
        }
    }
    __syncthreads();
    // 3c. scatter into the cells
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        if (tid + 256u * e < n) {
            const uint64_t key = mine[e];
            const uint32_t old = atomicAdd(&cell[cell_of((uint32_t)(key >> 32))], 1u);
            sk[(old >> 16) + (old & 0xffffu)] = key;
        }
    }
    __syncthreads();
    // 4. rank inside the cell, write the outputs
    for (uint32_t p = tid; p < n; p += 256) {
        const uint64_t key = sk[p];
        const uint32_t c = cell[cell_of((uint32_t)(key >> 32))];
        const uint32_t start = c >> 16, m = c & 0xffffu;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; j++) rank += sk[start + j] < key ? 1u : 0u;
        const uint32_t pos = rg.x + start + rank;
        const uint32_t slot = (uint32_t)key;
        keys[pos] = ((uint64_t)tile << 32) | (key >> 32);
        slot_of[pos] = slot;
        point_list[pos] = slot_vp[slot];
    }
}

template <int BITS>
void sort_pass(hipStream_t s, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, uint32_t n,
               int shift, int width, uint32_t* hist, uint32_t nblk, const uint32_t* n_dev)
{
    constexpr int BINS = 1 << BITS;
    uint32_t* totals = hist + (size_t)BINS * nblk;
    const uint32_t mask = (1u << width) - 1u;
    hipLaunchKernelGGL(radix_hist_kernel<BITS>, dim3(nblk), dim3(256), 0, s, kin, n, shift, mask, hist, nblk, n_dev);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(BINS), dim3(256), 0, s, hist, nblk, totals);
    hipLaunchKernelGGL(radix_scatter_kernel<BITS>, dim3(nblk), dim3(256), 0, s, kin, vin, kout, vout, n, shift, mask,
                       hist, totals, nblk, n_dev);
}

}  // namespace

void launch_tile_order(hipStream_t s, const uint2* ranges, uint32_t tiles_total, uint32_t* perm)
{
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, ranges, tiles_total, perm);
}

void launch_tile_count(hipStream_t s, int VP, int P, const int* radii, GeomState g, uint2* ranges, uint32_t tiles_total,
                       int tiles_x, int tiles_y)
{
    (void)hipMemsetAsync(ranges, 0, (size_t)tiles_total * sizeof(uint2), s);
    const uint32_t nblk = (uint32_t)((VP + kAggThreads - 1) / kAggThreads);
    // the LDS window needs a workgroup's 1024 Gaussians inside two consecutive views and two views' tiles inside the window
    if (2u * tiles_x * tiles_y <= kAggWindow && (uint32_t)P >= kAggThreads)
        hipLaunchKernelGGL(tile_count_kernel<true>, dim3(nblk), dim3(kAggThreads), 0, s, VP, P, radii, g, ranges, (uint32_t)tiles_x,
                           (uint32_t)tiles_y);
    else
        hipLaunchKernelGGL(tile_count_kernel<false>, dim3(nblk), dim3(kAggThreads), 0, s, VP, P, radii, g, ranges, (uint32_t)tiles_x,
                           (uint32_t)tiles_y);
}

void launch_tile_scan(hipStream_t s, uint2* ranges, uint32_t tiles_total, uint32_t* cursor, uint32_t* stats, uint32_t* info)
{
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, ranges, tiles_total, cursor, stats, info);
}

void launch_tile_scatter(hipStream_t s, int VP, int P, const int* radii, GeomState g, uint64_t* bucket_keys, uint32_t* slot_vp,
                         uint32_t* cursor, int tiles_x, int tiles_y, const uint32_t* info)
{
    const uint32_t nblk = (uint32_t)((VP + kAggThreads - 1) / kAggThreads);
    if (2u * tiles_x * tiles_y <= kAggWindow && (uint32_t)P >= kAggThreads)
        hipLaunchKernelGGL(tile_scatter_kernel<true>, dim3(nblk), dim3(kAggThreads), 0, s, VP, P, radii, g, bucket_keys, slot_vp,
                           cursor, (uint32_t)tiles_x, (uint32_t)tiles_y, info);
    else
        hipLaunchKernelGGL(tile_scatter_kernel<false>, dim3(nblk), dim3(kAggThreads), 0, s, VP, P, radii, g, bucket_keys, slot_vp,
                           cursor, (uint32_t)tiles_x, (uint32_t)tiles_y, info);
}

void launch_tile_sort(hipStream_t s, const uint2* ranges, uint32_t tiles_total, const uint64_t* bucket_keys, uint64_t* keys,
                      uint32_t* point_list, uint32_t* slot_of, const uint32_t* slot_vp)
{
    hipLaunchKernelGGL(tile_sort_kernel, dim3(tiles_total), dim3(256), 0, s, ranges, bucket_keys, keys, point_list, slot_of,
                       slot_vp);
}

void launch_scan_block_sums(hipStream_t s, uint32_t* block_sums, uint32_t nblocks, uint32_t* info, uint32_t capacity)
{
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nblocks, info, capacity);
}

void launch_duplicate(hipStream_t s, int VP, int P, const int* radii, GeomState g, uint64_t* keys_out,
                      uint32_t* vals_out, uint32_t* slot_vp, uint4* rowpos, int tiles_x, int tiles_y, const uint32_t* info)
{
    const uint32_t nblk = (uint32_t)((VP + kGaussBlock - 1) / kGaussBlock);
    hipLaunchKernelGGL(duplicate_kernel, dim3(nblk), dim3(kGaussBlock), 0, s, VP, P, radii, g, keys_out, vals_out,
                       slot_vp, rowpos, (uint32_t)tiles_x, (uint32_t)tiles_y, info);
}

// Sorts (keys, vals) of length R on the low plan.total_bits bits.  The unsorted input sits in
// the *_alt buffers when start_in_alt, else in the primary buffers; the caller picks that so
// the final pass lands in b.keys / b.point_list.
// n_dev != NULL (sync-free form): R is the CAPACITY the grids and the histogram layout are sized for, the live key count is
// read from *n_dev by every kernel.
void launch_radix_sort(hipStream_t s, BinningState b, uint32_t R, SortPlan plan, bool start_in_alt, const uint32_t* n_dev)
{
    if (R == 0) return;
    const uint32_t nblk = (R + kSortTile - 1) / kSortTile;
    uint64_t* k[2] = {b.keys, b.keys_alt};
    uint32_t* v[2] = {b.point_list, b.point_list_alt};
    int cur = start_in_alt ? 1 : 0;
    for (int p = 0; p < plan.passes; p++) {
        const int shift = p * plan.digit_bits;
        int width = plan.live_bits - shift;
        if (width > plan.digit_bits) width = plan.digit_bits;
        switch (plan.digit_bits) {
            case 8: sort_pass<8>(s, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], R, shift, width, b.sort_hist, nblk, n_dev); break;
            case 9: sort_pass<9>(s, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], R, shift, width, b.sort_hist, nblk, n_dev); break;
            default: sort_pass<10>(s, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], R, shift, width, b.sort_hist, nblk, n_dev); break;   // Morton codes (raster_scene.hip)
        }
        cur ^= 1;
    }
}

void launch_tile_ranges(hipStream_t s, const uint64_t* keys, uint32_t R, uint2* ranges, uint32_t tiles_total,
                        uint32_t* point_list, const uint32_t* slot_vp, uint32_t* slot_of, const uint32_t* n_dev)
{
    (void)hipMemsetAsync(ranges, 0, (size_t)tiles_total * sizeof(uint2), s);
    if (R > 0)
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, keys, R, ranges, point_list,
                           slot_vp, slot_of, n_dev);
}

}  // namespace gd
