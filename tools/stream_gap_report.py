#!/usr/bin/env python3
"""GPU idle time between the kernels of a streaming chunk, from a rocprofv3 --kernel-trace CSV of tools/bench_stream.py: per chunk (kernels between two
`mel_logmel_kernel` launches) the span, the sum of the kernel durations and the gaps -- what a hipGraph of the chunk's launch chain could recover at most.
usage: python tools/stream_gap_report.py <kt_kernel_trace.csv>"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    chunks, cur = [], []
    for s, e, n in rows:
        if "mel_logmel" in n and cur:
            chunks.append(cur); cur = []
        cur.append((s, e, n))
    chunks = [c for c in chunks if len(c) > 150][5:]                 # steady state
    if not chunks:
        print("no steady-state chunks"); return
    tot_span = tot_busy = 0.0
    gaps_all = []
    for c in chunks:
        span = (c[-1][1] - c[0][0]) / 1e3
        busy = sum(e - s for s, e, _ in c) / 1e3
        gaps = [(c[i + 1][0] - c[i][1]) / 1e3 for i in range(len(c) - 1)]
        tot_span += span; tot_busy += busy; gaps_all += gaps
    n = len(chunks)
    gaps_all.sort()
    big = [g for g in gaps_all if g > 5.0]
    print(f"{n} chunks, {len(chunks[0])} kernels each: first kernel start -> last kernel end {tot_span / n:.1f} us, kernels' own time {tot_busy / n:.1f} us, "
          f"idle between kernels {(tot_span - tot_busy) / n:.1f} us per chunk")
    print(f"gap between consecutive kernels: median {gaps_all[len(gaps_all) // 2]:.2f} us, p90 {gaps_all[len(gaps_all) * 9 // 10]:.2f}, "
          f"gaps > 5 us: {len(big) / n:.1f} per chunk, {sum(big) / n:.1f} us per chunk")
    # where the large gaps sit: the kernel that FOLLOWS them
    where = {}
    for c in chunks:
        for i in range(len(c) - 1):
            g = (c[i + 1][0] - c[i][1]) / 1e3
            if g > 5.0:
                k = c[i + 1][2].split("<")[0].split("(")[0][-40:]
                where[k] = where.get(k, 0.0) + g
    for k, v in sorted(where.items(), key=lambda kv: -kv[1])[:8]:
        print(f"   {v / n:7.1f} us per chunk in front of {k}")


if __name__ == "__main__":
    main()
