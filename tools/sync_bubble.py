#!/usr/bin/env python
"""GPU idle time around the rasterizer forward's host sync, from a rocprofv3 kernel trace: per step, the gaps between
consecutive kernels from scan_block_sums_kernel (last kernel before the num_rendered read-back) to the first kernel of
the VAE graph (conv3x3_first_kernel).  Upper bound: rocprofv3 inflates host launch costs.
usage: sync_bubble.py <kernel_trace.csv>"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
starts = [i for i, r in enumerate(rows) if "scan_block_sums_kernel" in name(r)]
res = []
for i in starts:
    j = i
    while j < len(rows) and "conv3x3_first_kernel" not in name(rows[j]):
        j += 1
        if j - i > 400:
            break
    if j >= len(rows) or j - i > 400:
        continue
    seg = rows[i:j + 1]
    gaps = [max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(seg[:-1], seg[1:])]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg[1:-1])
    res.append((gaps[0] / 1e3, sum(gaps) / 1e3, busy / 1e3, len(seg) - 2))
res = res[3:]
print(f"{len(res)} steps: sync gap (scan -> duplicate) median {statistics.median(r[0] for r in res):.1f} us; "
      f"all gaps scan -> first VAE kernel median {statistics.median(r[1] for r in res):.1f} us; "
      f"kernel time in between {statistics.median(r[2] for r in res):.1f} us over {res[0][3]} kernels")
