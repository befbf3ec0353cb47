"""Design study for the backward blend: replay the oracle's tile lists of the benchmark scene in numpy and measure, for
candidate pixel-block shapes, how dense the (entry, pixel) contribution matrix of a block is.  CPU only (uses the
oracle: tools/, not product)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from tests import helpers as h

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
inp = h.raster_inputs(P=P, H=HW, W=HW, seed=0, azimuth=0.0)
st = h.oracle_forward(inp)
print("R", st.num_rendered, "visited", st.pairs_visited_fwd, "blended", st.pairs_blended_fwd)
gx = HW // 16
rng = np.random.default_rng(0)
tiles = rng.choice(gx * gx, size=min(gx * gx, 160), replace=False)
py, px = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
shapes = {"16x4": (16, 4), "8x8": (8, 8), "4x4": (4, 4), "8x4": (8, 4), "8x2": (8, 2), "16x1": (16, 1), "16x16": (16, 16)}
acc = {k: [0, 0, 0] for k in shapes}   # records, (entry, block) pairs with >=1 record, blocks
chunk_stats = {16: [0, 0], 32: [0, 0], 64: [0, 0]}
strip_iters = {"4x4rows": 0, "strip64": 0, "strips": 0}
for t in tiles:
    ty, tx = divmod(int(t), gx)
    a, b = st.ranges[t]
    ids = st.point_list[a:b]
    n = len(ids)
    if n == 0:
        continue
    xy = st.means2D[ids]; co = st.conic_opacity[ids]
    X = (tx * 16 + px).astype(np.float32); Y = (ty * 16 + py).astype(np.float32)
    T = np.ones((16, 16), np.float32); done = np.zeros((16, 16), bool)
    contrib = np.zeros((n, 16, 16), bool)
    for j in range(n):
        dx = xy[j, 0] - X; dy = xy[j, 1] - Y
        power = -0.5 * (co[j, 0] * dx * dx + co[j, 2] * dy * dy) - co[j, 1] * dx * dy
        alpha = np.minimum(0.99, co[j, 3] * np.exp(power))
        ok = (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
        testT = T * (1 - alpha)
        stop = ok & (testT < 1e-4)
        done |= stop
        ok &= ~stop
        contrib[j] = ok
        T = np.where(ok, testT, T)
        if done.all():
            break
    for k, (bw, bh) in shapes.items():
        c = contrib.reshape(n, 16 // bh, bh, 16 // bw, bw).sum(axis=(2, 4))   # [n, by, bx]
        acc[k][0] += int(c.sum()); acc[k][1] += int((c > 0).sum()); acc[k][2] += c.shape[1] * c.shape[2]
    # lockstep iteration model: strip = 16x4, rows = its four 4x4 blocks, chunks of 16 entries per block per group of 64
    c44 = contrib.reshape(n, 4, 4, 4, 4).sum(axis=(2, 4))    # [n, by(4), bx(4)]
    for s in range(4):
        nz_strip = np.nonzero(c44[:, s, :].sum(axis=1) > 0)[0]
        strip_iters["strips"] += 1
        m = len(nz_strip)
        strip_iters["strip64"] += 64 * ((m + 63) // 64)
        nball = (c44[nz_strip, s, :] > 0).sum(axis=0)
        strip_iters.setdefault("whole", 0); strip_iters["whole"] += 16 * int(np.ceil(nball.max() / 16))
        strip_iters.setdefault("sumblk", 0); strip_iters["sumblk"] += 16 * int(np.ceil(nball / 16).sum())
        strip_iters.setdefault("nent", 0); strip_iters["nent"] += m
        for g0 in range(0, m, 64):
            grp = nz_strip[g0:g0 + 64]
            nb = (c44[grp, s, :] > 0).sum(axis=0)          # entries per block in this group
            strip_iters["4x4rows"] += 16 * int(np.ceil(nb.max() / 16))
recs = acc["4x4"][0]
for k, v in acc.items():
    bw, bh = shapes[k]
    print(f"{k:6s} records {v[0]:9d}  (entry,block) pairs {v[1]:9d}  density {v[0] / (v[1] * bw * bh):.3f}  entries/block {v[1] / v[2]:.1f}")
print("strips", strip_iters["strips"], "iters/strip: 4x4rows", strip_iters["4x4rows"] / strip_iters["strips"],
      " strip64", strip_iters["strip64"] / strip_iters["strips"],
      " lane efficiency 4x4rows", recs / (strip_iters["4x4rows"] * 64.0), " strip64", recs / (strip_iters["strip64"] * 64.0))

print("whole-list chunking: iters/strip", strip_iters["whole"] / strip_iters["strips"], "eff", recs / (strip_iters["whole"] * 64.0))
print("independent block rows (sum of chunks over blocks x16 lanes): eff", recs / (strip_iters["sumblk"] * 16.0))
print("entries with records per strip", strip_iters["nent"] / strip_iters["strips"])
