#!/usr/bin/env python3
"""Per-kernel weight-stream rate of a streaming chunk: reads a rocprofv3 kernel table written by tools/rocprof_summary.py for
`tools/bench_stream.py [--bf16]` on nemotron-600m (d 1024, ffn 4096, 24 layers) and prints, for every product kernel, the weight bytes one launch
streams from HBM and the rate that makes of its average duration -- the table the round-4 verdict asked for (item 3).
usage: python tools/stream_kernel_tbps.py profiles/r05_m3_stream_bf16_kernel_stats.md [bf16|fp32] > profiles/r05_m3_stream_bf16_kernel_tbps.md"""
import re
import sys

D, F = 1024, 4096


def product_of(name, mode):
    """(label, weight elements) of a small-M product kernel by its template arguments; None for other kernels."""
    m = re.match(r"`gemm_smallm_bf16_kernel<([^>]*)>`", name)
    if m and mode == "bf16":
        a = [x.strip() for x in m.group(1).split(",")]
        epi, steps, a16, ln, ct = int(a[0]), int(a[1]), a[3] == "true", a[4] == "true", int(a[5])
        pre = len(a) > 11 and a[11] == "true"
        if epi == 3 and a16: return "ffn fc2 + residual (K 4096)", D * F
        if epi == 3: return "attention out_proj / conv pw2 + residual", D * D
        if epi == 4: return "conv pw1 + GLU (+ depthwise conv tail)", 2 * D * D
        if epi == 2 and ln and ct == 2: return "ffn fc1 + SiLU, LayerNorm folded" + (" (+ the previous block's final norm)" if pre else ""), D * F
        if epi == 0 and ln and ct == 2: return "attention qkv, LayerNorm folded", 3 * D * D
        return None
    m = re.match(r"`gemm_smallm_(ln_)?kernel<([^>]*)>`", name)
    if m and mode == "fp32":
        a = [x.strip() for x in m.group(2).split(",")]
        epi = int(a[0])
        if m.group(1):
            return {2: ("ffn fc1 + SiLU, LayerNorm folded", D * F), 4: ("conv pw1 + GLU (+ conv tail), LayerNorm folded", 2 * D * D),
                    0: ("attention qkv, LayerNorm folded", 3 * D * D)}.get(epi)
        if epi == 3: return "residual products: ffn fc2 (K 4096, 2 of 4 launches) and out_proj / pw2 (K 1024)", (D * F + D * D) // 2
    return None


def main():
    path, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16")
    wbytes = 2 if mode == "bf16" else 4
    print(f"# weight-stream rate per product kernel, nemotron-600m streaming, 16 sessions x 160 ms chunks, {mode} weights; from `{path}`")
    print("# rate = weight bytes of one launch / average duration (activations, launch ramp and the product itself are inside the duration)\n")
    print("| kernel | product | launches | avg us | weight MB / launch | TB/s |\n|---|---|---:|---:|---:|---:|")
    tot_b, tot_us = 0.0, 0.0
    for line in open(path):
        c = [x.strip() for x in line.split("|")]
        if len(c) < 8 or not c[2].isdigit():
            continue
        p = product_of(c[1], mode)
        if not p:
            continue
        n, avg = int(c[2]), float(c[4])
        mb = p[1] * wbytes / 1e6
        print(f"| {c[1]} | {p[0]} | {n} | {avg:.2f} | {mb:.2f} | {mb / avg:.2f} |")
        tot_b += n * mb; tot_us += n * avg
    print(f"\nall product launches together: {tot_b / tot_us:.2f} TB/s of weights over their own durations")


if __name__ == "__main__":
    main()
