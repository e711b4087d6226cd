#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2).
Units / gfx950 correction (same guide, section HBM): both counters are in KiB; FETCH_SIZE tallies 128-byte requests of wide
coalesced reads at 64 B, so it is DOUBLED; WRITE_SIZE is taken as is and sanity-checked here against the fc1 GEMM,
whose output (M x N fp32) is a known byte count.
usage: pmc_hbm.py <fetch_dir> <write_dir> <out.json> [fc1-kernel-prefix [M N]]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone .*?\]", "", name)
    m = re.match(r"(?:void )?(?:pk::)?([A-Za-z0-9_]+)(<[^(]*>)?", name.strip())
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def load(d, counter):
    tot, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return {k: tot[k] / max(1, len(n[k])) for k in tot}, {k: len(v) for k, v in n.items()}


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch; read = 2 * FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB", "kernels": {}}
    for k in sorted(fetch, key=lambda k: -(fetch[k] * 2 + write.get(k, 0))):
        rd, wr = 2.0 * fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"launches_seen": nf[k], "read_bytes": round(rd), "write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
    prefix = sys.argv[4] if len(sys.argv) > 4 else "gemm_pipe_kernel<4, 2, 1, 2, 32, 2"                  # ffn fc1 + SiLU (EPI_SILU = 2)
    out_bytes = int(sys.argv[5]) * int(sys.argv[6]) * int(sys.argv[7]) if len(sys.argv) > 7 else 8064 * 2048 * 4   # M N bytes-per-element of its output
    fc1 = next((k for k in out["kernels"] if k.startswith(prefix)), None)
    if fc1:
        out["ffn_fc1_silu_kernel"] = fc1
        out["ffn_fc1_silu_bytes_per_launch"] = out["kernels"][fc1]["hbm_bytes"]
        out["ffn_fc1_silu_write_check"] = {"counter_bytes": out["kernels"][fc1]["write_bytes"], "algorithmic_bytes": out_bytes}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out["kernels"].items())[:14]:
        print(f"{k:48s} read {v['read_bytes']/1e6:9.2f} MB  write {v['write_bytes']/1e6:9.2f} MB")
    print({k: v for k, v in out.items() if k.startswith("ffn")})


if __name__ == "__main__":
    main()
