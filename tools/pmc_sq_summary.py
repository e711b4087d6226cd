#!/usr/bin/env python3
"""Per-kernel SQ counter table from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE): mean per dispatch.
'us @2.4 GHz' = GRBM_GUI_ACTIVE / 8 / 2.4 GHz: against the kernel's measured duration it gives the effective clock of the (profiled) run.
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs).  usage: pmc_sq_summary.py <dir> <out.md> [title]"""
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
    agg, disp = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    rows = []
    for k, c in agg.items():
        n = len(disp[k])
        g = c.get("GRBM_GUI_ACTIVE", 0.0) / n
        busy, wave = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n, c.get("SQ_WAVE_CYCLES", 0.0) / n
        util = busy / (g / 8 * 256 * 4) if g else 0.0
        rows.append((g * n, k, n, util, g / 8 / 2400.0, busy, wave, c.get("SQ_WAIT_ANY", 0) / n / max(wave, 1), c.get("SQ_WAIT_INST_ANY", 0) / n / max(wave, 1),
                     c.get("SQ_ACTIVE_INST_VALU", 0) / n / max(wave, 1), c.get("SQ_LDS_BANK_CONFLICT", 0) / n, c.get("SQ_INSTS_LDS", 0) / n,
                     c.get("SQ_WAIT_INST_LDS", 0) / n / max(wave, 1)))
    rows.sort(reverse=True)
    out = [sys.argv[3] if len(sys.argv) > 3 else "# rocprofv3 --pmc, SQ counters, mean per dispatch", "",
           "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); us = GRBM_GUI_ACTIVE / 8 / 2.4 GHz; the SQ_WAIT_* / "
           "SQ_ACTIVE_* ratios are per SQ_WAVE_CYCLES (quad-cycles summed over waves).", "",
           "| kernel | dispatches | MfmaUtil | us @2.4 GHz | MFMA busy cyc | wave cyc | wait_any / wave | wait_inst / wave | wait_inst_lds / wave | VALU active / wave | LDS bank conflict cyc | LDS insts |",
           "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for _, k, n, util, us, busy, wave, wa, wi, va, lc, li, wl in rows[:28]:
        out.append(f"| `{k}` | {n} | {util:.3f} | {us:.1f} | {busy:.3g} | {wave:.3g} | {wa:.2f} | {wi:.2f} | {wl:.3f} | {va:.2f} | {lc:.3g} | {li:.3g} |")
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main()
