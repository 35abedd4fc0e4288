#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into text (kernel durations + PMC sums per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, kernel_sub="fused"):
    lines = []
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"== kernel stats ({os.path.relpath(f, d)})")
        for r in csv.DictReader(open(f)):
            lines.append("  {Name:60.60s} calls={Calls:>5s} avg_ns={AverageNs:>12s} min_ns={MinNs:>10s} max_ns={MaxNs:>10s} pct={Percentage}".format(**r))
    for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(p):
            continue
        for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            n = defaultdict(set)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[k].add(r["Dispatch_Id"])
            lines.append(f"== {os.path.basename(p)}  (per-dispatch mean)")
            for k in acc:
                if kernel_sub not in k:
                    continue
                nd = max(1, len(n[k]))
                vg = ""
                lines.append(f"  {k[:70]}  dispatches={nd} {vg}")
                for c, v in sorted(acc[k].items()):
                    lines.append(f"      {c:28s} {v / nd:18.1f}")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
