#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into small per-kernel summaries (the raw traces are too big to keep).

usage: summarize_prof.py <rocprof_dir> <prefix> <out_txt>
 - <prefix>_kernel_stats.csv is copied as is (already a summary);
 - <prefix>_counter_collection.csv is aggregated: per (kernel name, counter) -> dispatches, mean, sum."""
import csv
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][-60:]


def main():
    d, prefix, out = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = []
    ks = os.path.join(d, prefix + "_kernel_stats.csv")
    if os.path.exists(ks):
        lines.append("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
        for row in csv.reader(open(ks)):
            lines.append(",".join(row))
    cc = os.path.join(d, prefix + "_counter_collection.csv")
    if os.path.exists(cc):
        agg = defaultdict(lambda: [0, 0.0])
        rd = csv.DictReader(open(cc))
        for r in rd:
            key = (short(r.get("Kernel_Name", "")), r.get("Counter_Name", ""))
            a = agg[key]
            a[0] += 1
            a[1] += float(r.get("Counter_Value", 0) or 0)
        lines.append("== PMC counters per kernel (sum over dispatches / dispatches) ==")
        lines.append("kernel,counter,dispatches,mean_per_dispatch,sum")
        for (k, c), (n, s) in sorted(agg.items()):
            lines.append(f"{k},{c},{n},{s / max(n, 1):.6g},{s:.6g}")
    kt = os.path.join(d, prefix + "_kernel_trace.csv")
    if os.path.exists(kt) and not os.path.exists(ks):
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(kt)):
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        lines.append("== kernel durations from the trace of this (counter) run: kernel,calls,avg_us ==")
        for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{k},{n},{s / n:.3f}")
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
