#!/usr/bin/env python
"""Per-kernel statistics of the STEADY-STATE part of a rocprofv3 --kernel-trace CSV.

rocprofv3 --stats aggregates the whole process, including MIOpen's first-call solver search in
the warm-up steps.  This tool keeps only kernels that start after the `skip`-th launch of a marker
kernel (default: the rasterizer's backward render kernel, launched once per bench step) and writes
the same columns as rocprofv3's *_kernel_stats.csv.

    python tools/steady_stats.py <kernel_trace.csv> <out.csv> [--marker NAME] [--skip N]
"""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("out")
    ap.add_argument("--marker", default="render_backward_block_kernel")
    ap.add_argument("--skip", type=int, default=2)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["End_Timestamp"]) for r in rows if a.marker in r["Kernel_Name"]]
    t0 = marks[a.skip - 1] if a.skip > 0 and len(marks) >= a.skip else 0
    steps = max(len(marks) - a.skip, 1)
    agg = collections.OrderedDict()
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < t0:
            continue
        d = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        d[0] += 1
        d[1] += e - s
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
    tot = sum(d[1] for d in agg.values()) or 1
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsPerStep",
                    "MsPerStep"])
        for name, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, d[0], d[1], d[1] / d[0], 100.0 * d[1] / tot, d[2], d[3], d[0] / steps,
                        d[1] / steps / 1e6])
    print(f"steady window: {steps} step(s), {tot / 1e6 / steps:.3f} ms of kernel time per step, {len(agg)} kernels")


if __name__ == "__main__":
    main()
