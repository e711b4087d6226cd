#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (counter_collection csv) per kernel: mean counter value per dispatch.
usage: rocprof_pmc_summary.py <dir-or-csv> [out.md]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone .*?\]", "", name)
    m = re.match(r"(?:void )?(?:pk::)?([A-Za-z0-9_]+)(<[^(]*>)?", name.strip())
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    path = sys.argv[1]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    counters = sorted({c for v in agg.values() for c in v})
    lines = ["| kernel | dispatches | " + " | ".join(counters) + " |", "|---|---:|" + "---:|" * len(counters)]
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get(counters[0], 0))):
        n = max(1, len(disp[k]))
        lines.append(f"| `{k}` | {n} | " + " | ".join(f"{agg[k].get(c, 0) / n:.4g}" for c in counters) + " |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write("Mean counter value per dispatch.\n\n" + out + "\n")


if __name__ == "__main__":
    main()
