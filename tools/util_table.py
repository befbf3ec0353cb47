#!/usr/bin/env python
"""Markdown table of matrix-pipe utilisation and HBM bandwidth per kernel from a profiles/rNN_pmc.json (tools/pmc_all.sh):
    python tools/util_table.py profiles/r05_pmc.json [min_share_percent]
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)   (0.976 on an MFMA-only stream,
profiles/r05_mfma_util_check.txt); hbm GB/s = (2 x FETCH_SIZE + WRITE_SIZE) KiB / duration; clock = GUI cycles per XCD / duration.
All three are taken under the profiler (clocks run ~3 % lower there than in an unprofiled run)."""
import json
import sys

d = json.load(open(sys.argv[1]))
min_share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ks = d["kernels"]
tot = sum(v.get("avg_duration_us_profiled", 0.0) * v.get("launches_sampled", 0) for v in ks.values())
rows = []
for k, v in ks.items():
    t = v.get("avg_duration_us_profiled", 0.0) * v.get("launches_sampled", 0)
    if tot <= 0 or 100.0 * t / tot < min_share:
        continue
    rows.append((t, k, v))
rows.sort(reverse=True)
print(f"kernel-source hash {d.get('kernel_source_hash')}, workload {d.get('workload')}")
print("| kernel | launches sampled | avg us | share of sampled kernel time | mfma_util | HBM GB/s | of 8 TB/s | clock GHz |")
print("|---|---|---|---|---|---|---|---|")
for t, k, v in rows:
    mu = v.get("mfma_util")
    gb = v.get("hbm_gbps")
    print(f"| `{k[:70]}` | {v.get('launches_sampled', 0)} | {v.get('avg_duration_us_profiled', 0.0):.1f} | {100.0 * t / tot:.1f} % | "
          f"{'%.3f' % mu if mu is not None else '-'} | {'%.0f' % gb if gb is not None else '-'} | "
          f"{'%.3f' % (gb / 8000.0) if gb is not None else '-'} | {'%.2f' % v['clock_ghz'] if 'clock_ghz' in v else '-'} |")
