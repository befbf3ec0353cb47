"""Measured parity errors -> ``gpurun_out/parity_report.json`` (merged back from the GPU box by gpurun and copied
into ``profiles/`` by hand).  Pure bookkeeping: every test asserts its own tolerance; this file only remembers
the largest error each check observed, so that BASELINE.md's tolerance table can quote measured numbers."""
from __future__ import annotations

import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "gpurun_out", "parity_report.json")


def record(case: str, name: str, **numbers) -> None:
    try:
        os.makedirs(os.path.dirname(_PATH), exist_ok=True)
        data = json.load(open(_PATH)) if os.path.exists(_PATH) else {}
        entry = data.setdefault(case, {}).setdefault(name, {})
        for k, v in numbers.items():
            v = float(v)
            entry[k] = max(entry.get(k, 0.0), v) if k.startswith("max") else v
        json.dump(data, open(_PATH, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
