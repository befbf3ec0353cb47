"""``bench.py --gpus N`` launches N ranks itself and refuses a world that is not N (CPU, gloo, stubbed step).

VERDICT r03 weak #8: the flag used to be parsed and ignored, so ``python bench.py --gpus 8`` measured one GPU under an
8-GPU label.  The step is stubbed (``--stub-step``: one small all-reduce) because the real one needs an MI355X; what is
under test is the launcher, the world-size check and the rank accounting that the real line shares.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_gpus_2_without_torchrun_launches_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    assert line["steps"] == 3 and line["warmup"] == 1
    assert line["metric"] == "stub"          # never to be mistaken for a measurement


def test_world_size_mismatch_is_refused():
    # an external launcher that started ONE rank for --gpus 2: refused, nothing printed on stdout
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-step"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "WORLD_SIZE=1" in r.stderr


def test_single_rank_stub_line():
    r = _run(["--steps", "2", "--warmup", "0", "--stub-step"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1


import pytest  # noqa: E402


@pytest.mark.gpu
def test_the_real_step_under_the_launcher_two_ranks_share_one_gpu_over_gloo():
    """The REAL ``bench.py --gpus 2 --steps 2`` (round-4 review: the stub covers the launcher, not the step under it): the
    process re-executes itself under torch.distributed.run with two ranks; GD_DIST_BACKEND=gloo lets both share the one GPU
    of the test box (each renders its 4 of the 8 views with the full-size nets, gradients all-reduced through gloo).  One
    JSON line, two ranks accounted for, four views per rank, a finite positive rate, a healthy scene."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "2", "--no-cpu-baseline"], {"GD_DIST_BACKEND": "gloo"}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo"
    assert line["config"]["views_per_gpu"] == 4 and line["config"]["views"] == 8
    assert line["steps"] == 2 and line["value"] > 0 and line["ms_per_step"] > 0
    assert line["health"]["params_finite"] and line["health"]["visible_after_timed_steps"] > 0
    assert line["config"]["batch_invariant"] is False and line["config"]["library_fallbacks"] == 0
    assert "cpu_baseline" not in line and line["metric"].startswith("SDS iters/sec")
    # the one data-path collective is timed inside the timed region (HIP events either side of it) and sized
    ar = line["config"]["grad_allreduce"]
    assert ar["calls_per_step"] == 1 and ar["ms_per_step"] > 0 and ar["bytes"] >= 100000 * 14 * 4
