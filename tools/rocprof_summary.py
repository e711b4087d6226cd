#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or kernel_trace csv) into the per-kernel table
committed under profiles/: calls, total / average / min / max duration.  usage: rocprof_summary.py <db|csv> [out.md]"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone .*?\]", "", name)
    m = re.match(r"(?:void )?(?:pk::)?([A-Za-z0-9_]+)(<[^(]*>)?", name.strip())
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def load(path):
    rows = []
    if path.endswith(".db"):
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        ncol = "name" if "name" in cols else "kernel_name"
        s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        for n, a, b in db.execute(f"select {ncol}, {s}, {e} from kernels"):
            rows.append((n, (b - a) / 1000.0))
    else:
        for r in csv.DictReader(open(path)):
            rows.append((r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0))
    return rows


def main():
    rows = load(sys.argv[1])
    agg = defaultdict(list)
    for n, us in rows:
        agg[short(n)].append(us)
    tot = sum(sum(v) for v in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| `{k}` | {len(v)} | {sum(v) / 1000:.3f} | {sum(v) / len(v):.2f} | {min(v):.2f} | {max(v):.2f} | {100 * sum(v) / tot:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
