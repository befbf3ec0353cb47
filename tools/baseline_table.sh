#!/bin/bash
# Numbers for BASELINE.md section 4 (configs C1..C5 on one MI355X box) -> gpurun_out/baseline_table.txt
out=gpurun_out/baseline_table.txt; mkdir -p gpurun_out; : > $out
python - >> $out 2>/dev/null <<'PY'
import time, numpy as np
from tests import helpers as h
from oracle import gd_oracle
for name, P, HW in (("C1", 10000, 256), ("C2", 100000, 512)):
    inp = h.raster_inputs(P=P, H=HW, W=HW, seed=0)
    a = (inp["bg"], inp["means3D"], inp["colors_precomp"], inp["opacities"], inp["scales"], inp["rotations"], inp["scale_modifier"], inp["cov3D_precomp"], inp["viewmatrix"], inp["projmatrix"], inp["tanfovx"], inp["tanfovy"], inp["image_height"], inp["image_width"], inp["sh"], inp["degree"], inp["campos"])
    gc, gd, ga = h.random_image_grads(HW, HW)
    for omp in (False, True):
        t0 = time.perf_counter(); st = gd_oracle.forward(*a, omp=omp); t1 = time.perf_counter(); gd_oracle.backward(st, gc, gd, ga); t2 = time.perf_counter()
        print(f"{name} CPU oracle {'OpenMP' if omp else '1 thread'}: fwd {1e3*(t1-t0):.1f} ms bwd {1e3*(t2-t1):.1f} ms  (P={P}, {HW}^2, R={st.num_rendered})")
PY
python bench.py --raster-only --views 1 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['raster_kernels_ms_per_step']
print('C2 MI355X 1 view 100k/512^2: forward chain %.3f ms (preprocess+scan+duplicate+sort+ranges+render_fwd), backward chain %.3f ms (render_bwd+preprocess_bwd); %s' % (sum(r[k] for k in ('preprocess','scan','duplicate','sort','ranges','render_fwd')), r['render_bwd'] + r['preprocess_bwd'], {k: round(v, 4) for k, v in r.items()}))" >> $out
python bench.py --views 4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('C3 MI355X 4 views: %.2f ms/step = %.2f iters/s; dense frac %.3f' % (d['ms_per_step'], d['value'], d['roofline_dense']['frac']))" >> $out
python bench.py --vsd --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('C5 MI355X VSD bf16 1 view: %.2f ms/step = %.2f view-iters/s; dense frac %.3f' % (d['ms_per_step'], d['value'], d['roofline_dense']['frac']))" >> $out
cat $out
