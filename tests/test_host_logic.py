"""Host-side logic that needs no GPU and no oracle."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_runtime_env_sets_graph_flag_before_hip_starts():
    """garmentdreamer_amd/_runtime_env.py: importing the package puts DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in place
    (fresh interpreter, no HIP call yet) and reports hipGraph replay as usable; an explicit user value wins."""
    import subprocess
    import sys as _sys
    code = ("import os, garmentdreamer_amd as g; "
            "print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), g._runtime_env.graph_replay_safe())")
    env = {k: v for k, v in os.environ.items() if k not in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "GD_HIP_GRAPHS_FORCE")}
    out = subprocess.run([_sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True)
    assert out.stdout.split() == ["0", "True"]
    env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
    out = subprocess.run([_sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True)
    assert out.stdout.split() == ["1", "False"]


def test_flat_adam_leaves_cpu_and_non_fp32_parameters_to_torch_adam():
    """flat_adam.FlatAdam on a machine without a GPU: nothing qualifies for the flat (HIP) set, every parameter is stepped
    by torch.optim.Adam with the same hyper-parameters -- bit for bit -- and zero_grad / lr changes reach it."""
    import torch
    from garmentdreamer_amd.flat_adam import FlatAdam
    g = torch.Generator().manual_seed(0)
    mine = [torch.nn.Parameter(torch.randn(4, 8, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g).to(torch.bfloat16))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt, ropt = FlatAdam(mine, lr=2e-3, betas=(0.8, 0.95), eps=1e-6, flat=mine), torch.optim.Adam(ref, lr=2e-3, betas=(0.8, 0.95), eps=1e-6)
    assert opt.flat_grad is None and not any(hasattr(p, "_gd_grad_sink") for p in mine)
    assert opt.param_groups[0]["lr"] == 2e-3 and len(opt.param_groups[0]["params"]) == 2
    for it in range(4):
        opt.zero_grad()
        ropt.zero_grad()
        assert all(p.grad is None for p in mine)
        for p, q in zip(mine, ref):
            gr = torch.randn(p.shape, generator=g).to(p.dtype)
            p.grad, q.grad = gr.clone(), gr.clone()
        if it == 2:
            for grp in opt.param_groups:         # the torch idiom reaches the inner optimizer
                grp["lr"] = 5e-4
            for grp in ropt.param_groups:
                grp["lr"] = 5e-4
        opt.step()
        ropt.step()
        assert all(torch.equal(p.detach(), q.detach()) for p, q in zip(mine, ref))
    import pytest
    with pytest.raises(ValueError):
        FlatAdam([])
    with pytest.raises(ValueError):              # `flat` names the always-written subset of `params`
        FlatAdam(mine, flat=[torch.nn.Parameter(torch.zeros(3))])
    assert FlatAdam(mine).flat_grad is None     # flat=None: nothing re-seated
