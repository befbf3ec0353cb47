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
