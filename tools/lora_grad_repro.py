#!/usr/bin/env python
"""Are the adapter gradients of the reduced-width VSD step reproducible from run to run?  Two runs of the SAME schedule (eager / eager,
graph / graph) on identical weights and inputs with UNTRAINED up-projections (N(0, 0.02) or N(0, 0.3)): cosine of all adapter gradients,
of the up- and of the down-projections.  (They are not: the up-projection gradients are uncorrelated between two identical runs -- a
token sum that cancels to the level of the library GEMMs' run-to-run differences; tests compare adapter gradients at a TRAINED state,
tests/test_configs_gpu.py::_trained_fp32_vsd.)   python tools/lora_grad_repro.py"""
import sys, torch
sys.path.insert(0, ".")
import garmentdreamer_amd  # noqa
from tests.test_configs_gpu import _vsd_objects, _vsd_step, _cos, DEV
kw_u = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
kw_v = dict(block_out_channels=(64, 64, 128, 128))
runs = {}
for name, graphs, std in (("eager a", False, 0.02), ("eager b", False, 0.02), ("graph a", True, 0.02), ("graph b", True, 0.02), ("eager big a", False, 0.3), ("eager big b", False, 0.3)):
    gd, lora, train, q = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=graphs)
    g = torch.Generator(DEV).manual_seed(77)
    with torch.no_grad():
        for layer in lora.lora_layers:
            for m in layer.values():
                m.up.weight.normal_(0, std, generator=g)
    runs[name] = [_vsd_step(gd, q, train, seed=60 + i) for i in range(3)]
    names = {id(p): n for n, p in lora.named_parameters()}
    idx = [i for i, p in enumerate(train) if "lora" in names[id(p)]]
    up = [i for i, p in enumerate(train) if names[id(p)].endswith("up.weight")]
    dn = [i for i, p in enumerate(train) if names[id(p)].endswith("down.weight")]
    del gd, lora, train, q
cat = lambda gr, ii: torch.cat([gr[i].flatten() for i in ii])
for a, b in (("eager a", "eager b"), ("graph a", "graph b"), ("eager a", "graph a"), ("eager big a", "eager big b")):
    for it in range(3):
        ga, gb = runs[a][it][3], runs[b][it][3]
        print(a, "|", b, it, "all %.4f up %.4f down %.4f  lu %.5f %.5f  |g| %.3e" % (_cos(cat(ga, idx), cat(gb, idx)), _cos(cat(ga, up), cat(gb, up)), _cos(cat(ga, dn), cat(gb, dn)), runs[a][it][2], runs[b][it][2], float(cat(ga, idx).norm())))
