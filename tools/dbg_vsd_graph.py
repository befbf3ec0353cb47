"""eager vs eager vs graphed LoRA gradients of the reduced-width VSD step (how reproducible are they run to run?)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import garmentdreamer_amd  # noqa
import test_configs_gpu as T

kw_u = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
kw_v = dict(block_out_channels=(64, 64, 128, 128))
runs = {}
for name, graphs in (("e1", False), ("e2", False), ("g", True)):
    gd, lora, train, q = T._vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=graphs)
    runs[name] = [T._vsd_step(gd, q, train, seed=sd) for sd in (9, 10)]
    pname = {id(p): n for n, p in lora.named_parameters()}
    names = [pname[id(p)] for p in train]
for a, b in (("e1", "e2"), ("e1", "g")):
    for it in range(2):
        ga, gb = runs[a][it][3], runs[b][it][3]
        keys = [i for i in ga if float(ga[i].abs().max()) > 0]
        c = T._cos(torch.cat([ga[i].flatten() for i in keys]), torch.cat([gb[i].flatten() for i in keys]))
        worst = sorted((T._cos(ga[i], gb[i]), names[i]) for i in keys)[:4]
        print(a, b, "iter", it, "cos all", c, "lu", runs[a][it][2], runs[b][it][2], "worst", worst)
