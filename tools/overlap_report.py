#!/usr/bin/env python3
"""How much longer do the encoder's kernels run when decode-loop kernels of the previous batch overlap them?  Reads a rocprofv3 --kernel-trace
csv (kt_kernel_trace.csv of tools/experiments/final_measure.sh) and, per encoder kernel type, compares the launches that overlapped at least
one decode kernel (skinny_gemm / tdt_decide) with those that did not.  usage: overlap_report.py <kernel_trace.csv>"""
import bisect
import csv
import sys

import numpy as np


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    dec = sorted(k for k in K if "skinny_gemm" in k[2] or "tdt_decide" in k[2])
    ds, de = np.array([d[0] for d in dec]), np.array([d[1] for d in dec])
    print(f"{len(dec)} decode kernels, mean duration {np.mean(de - ds) / 1e3:.1f} us")
    for pat, label in (("gemm_pipe_kernel<4, 2, 1, 2, 32, 2, 1>", "ffn fc1 + SiLU"), ("gemm_pipe_kernel<4, 2, 1, 2, 64, 3, 1>", "ffn fc2 + residual"),
                       ("gemm_pipe_kernel<4, 2, 1, 2, 32, 0, 1>", "qkv"), ("gemm_pipe_kernel<4, 2, 1, 2, 32, 4, 1>", "pw1 + GLU"), ("relpos_attention", "attention"),
                       ("layernorm_kernel", "layernorm")):
        X = []
        for s, e, n in K:
            if pat not in n:
                continue
            ov = cnt = 0
            for j in range(bisect.bisect_left(de, s), len(dec)):
                if ds[j] >= e:
                    break
                o = min(e, de[j]) - max(s, ds[j])
                if o > 0:
                    ov += o
                    cnt += 1
            X.append((e - s, ov, cnt))
        if not X:
            continue
        X = np.array(X, float)
        dur, cnt = X[:, 0] / 1e3, X[:, 2]
        no, yes = dur[cnt == 0], dur[cnt > 0]
        if len(no) and len(yes):
            print(f"{label:20s} alone: {len(no):4d} launches, {no.mean():7.1f} us | overlapping {cnt[cnt > 0].mean():.1f} decode kernels: {len(yes):4d} launches, "
                  f"{yes.mean():7.1f} us  (+{100 * (yes.mean() / no.mean() - 1):.0f} %)")


if __name__ == "__main__":
    main()
