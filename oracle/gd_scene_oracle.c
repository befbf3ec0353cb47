/*
 * gd_scene_oracle.c -- CPU restatement of the reference's simple-knn (TEST INFRASTRUCTURE ONLY: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline may use it; the product never does).
 *
 * Follows Garment_3DGS/gaussiansplatting/submodules/simple-knn/simple_knn.cu step by step:
 *   :190-197  bounding box by cub::DeviceReduce::Reduce with init = {0,0,0}  -> the origin is always inside
 *   :45-61    30-bit Morton code of ((c - min) / (max - min)) * 1023, truncated to uint32
 *   :204-211  stable radix sort of (code, index)
 *   :79-122   min / max of every run of BOX_SIZE = 1024 consecutive sorted points
 *   :153-186  per point: 3 best of the +-3 sorted neighbours -> reject radius; then every box whose distance
 *             is <= reject and <= current 3rd best is scanned; result (b0 + b1 + b2) / 3
 * Pinned by tests/test_scene_cpu.py against a brute-force 3-NN (numpy): the boxed search is exact, so both
 * must agree to the last bit when the squared distance uses the same expression order.  Build with
 * -ffp-contract=off (Makefile).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BOX_SIZE 1024

static uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

static void update3(const float* q, const float* p, float* best)
{
    const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    float dist = dx * dx + dy * dy + dz * dz;
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
    }
}

static float dist_box_point(const float* box, const float* p)
{
    float d[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++)
        if (p[k] < box[k] || p[k] > box[3 + k]) d[k] = fminf(fabsf(p[k] - box[k]), fabsf(p[k] - box[3 + k]));
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

/* points [P][3] -> mean_dists [P]; optionally codes_out [P] (Morton codes) and order_out [P] (sorted indices) */
void gdso_dist2(int P, const float* points, float* mean_dists, uint32_t* codes_out, uint32_t* order_out)
{
    if (P <= 0) return;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};               /* init = {0,0,0} */
    for (int i = 0; i < P; i++)
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(mn[k], points[3 * i + k]);
            mx[k] = fmaxf(mx[k], points[3 * i + k]);
        }
    uint32_t* codes = (uint32_t*)malloc(sizeof(uint32_t) * P);
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * P);
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * P);
    for (int i = 0; i < P; i++) {
        uint32_t c = 0;
        for (int k = 0; k < 3; k++) {
            const float t = ((points[3 * i + k] - mn[k]) / (mx[k] - mn[k])) * (float)((1 << 10) - 1);
            c |= prep_morton((uint32_t)t) << k;
        }
        codes[i] = c;
        idx[i] = (uint32_t)i;
    }
    /* stable LSD counting sort on the 30-bit code, 10 bits per pass */
    for (int pass = 0; pass < 3; pass++) {
        uint32_t count[1025];
        memset(count, 0, sizeof(count));
        for (int i = 0; i < P; i++) count[((codes[idx[i]] >> (10 * pass)) & 1023u) + 1]++;
        for (int d = 0; d < 1024; d++) count[d + 1] += count[d];
        for (int i = 0; i < P; i++) tmp[count[(codes[idx[i]] >> (10 * pass)) & 1023u]++] = idx[i];
        memcpy(idx, tmp, sizeof(uint32_t) * P);
    }
    const int nboxes = (P + BOX_SIZE - 1) / BOX_SIZE;
    float* boxes = (float*)malloc(sizeof(float) * 6 * nboxes);
    for (int b = 0; b < nboxes; b++) {
        float* bx = boxes + 6 * b;
        for (int k = 0; k < 3; k++) { bx[k] = FLT_MAX; bx[3 + k] = -FLT_MAX; }
        for (int i = b * BOX_SIZE; i < P && i < (b + 1) * BOX_SIZE; i++)
            for (int k = 0; k < 3; k++) {
                bx[k] = fminf(bx[k], points[3 * idx[i] + k]);
                bx[3 + k] = fmaxf(bx[3 + k], points[3 * idx[i] + k]);
            }
    }
    for (int i = 0; i < P; i++) {
        const float* q = points + 3 * idx[i];
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        const int lo = i - 3 < 0 ? 0 : i - 3, hi = i + 3 > P - 1 ? P - 1 : i + 3;
        for (int j = lo; j <= hi; j++)
            if (j != i) update3(q, points + 3 * idx[j], best);
        const float reject = best[2];
        best[0] = best[1] = best[2] = FLT_MAX;
        for (int b = 0; b < nboxes; b++) {
            const float dist = dist_box_point(boxes + 6 * b, q);
            if (dist > reject || dist > best[2]) continue;
            for (int j = b * BOX_SIZE; j < P && j < (b + 1) * BOX_SIZE; j++)
                if (j != i) update3(q, points + 3 * idx[j], best);
        }
        mean_dists[idx[i]] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    if (codes_out) memcpy(codes_out, codes, sizeof(uint32_t) * P);
    if (order_out) memcpy(order_out, idx, sizeof(uint32_t) * P);
    free(codes); free(idx); free(tmp); free(boxes);
}
