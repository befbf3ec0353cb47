"""Adapter gradients of the reduced-width VSD iteration: eager autograd vs (eager | graphs) x (FlatAdam sinks | plain .grad)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_configs_gpu import _vsd_objects, _vsd_step, _cos
from garmentdreamer_amd.flat_adam import FlatAdam
kw_u = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
kw_v = dict(block_out_channels=(64, 64, 128, 128))
seeds = (9, 10)
gd_e, _, train_e, q_e = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=False)
eager = [_vsd_step(gd_e, q_e, train_e, seed=sd) for sd in seeds]
def vec(g, keys): return torch.cat([g[i].flatten() for i in keys])
def report(tag, train, stepper):
    for (di_e, lat_e, lu_e, g_e), sd in zip(eager, seeds):
        di, lat, lu, g = stepper(sd)
        keys = [i for i in g_e if float(g_e[i].abs().max()) > 0 and train[i].dtype == torch.float32 and i in g]
        a, b = vec(g_e, keys), vec(g, keys)
        per = sorted(_cos(g_e[i], g[i]) for i in keys)
        print(f"{tag} seed {sd}: cos {_cos(a, b):.4f} norm ratio {float(b.norm() / a.norm()):.3f} per-tensor cos min {per[0]:.3f} "
              f"median {per[len(per) // 2]:.3f} max {per[-1]:.3f}; dL/dimage cos {_cos(di_e, di):.6f} lu {lu_e:.6f} {lu:.6f}")
# control: a second plain instance
gd, _, train, q = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=False)
report("control (second eager instance)", train, lambda sd: _vsd_step(gd, q, train, seed=sd))
# the SAME first instance again
report("first instance again", train_e, lambda sd: _vsd_step(gd_e, q_e, train_e, seed=sd))
# FlatAdam with everything excluded (no re-seating at all): .grad handling only
gd, _, train, q = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=False)
opt = FlatAdam(train, lr=0.0, exclude=train)
report("FlatAdam, all excluded", train, lambda sd: _vsd_step(gd, q, train, seed=sd, zero_grad=opt.zero_grad))
# re-seated parameters, plain .grad (set to None by the helper)
gd, _, train, q = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=False)
opt = FlatAdam(train, lr=0.0)
for p in train:
    if hasattr(p, "_gd_grad_sink"):
        del p._gd_grad_sink
report("re-seated, .grad = None each step", train, lambda sd: _vsd_step(gd, q, train, seed=sd))
