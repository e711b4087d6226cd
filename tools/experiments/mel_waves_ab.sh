#!/bin/bash
# tools/experiments/mel_waves_ab.sh -- frames (= wavefronts) per workgroup of mel_logmel_kernel: 4 (round 3) / 8 / 16 (round 4): kernel time in the bench
# run, parity of the timed batch, and the HBM write traffic of the kernel (rocprofv3 --pmc WRITE_SIZE, its own pass)
export TMPDIR=/tmp
out=gpurun_out/mel_waves_ab.txt
: > $out
for rep in 1 2; do
  for w in 12 16; do
    lib=parakeet.cpp_amd/libparakeet_amd_mel$w.so; [ $w = 16 ] && lib=parakeet.cpp_amd/libparakeet_amd.so
    line=$(PK_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "waves=$w $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f mel_logmel_ms=%.4f mel_normalize_ms=%.4f stage_mel=%.3f'%(d['ms_per_step'],k['mel_logmel']['ms'],k['mel_normalize']['ms'],d['stage_ms']['mel']))" "$line")" >> $out
  done
done
for w in 16; do
  lib=parakeet.cpp_amd/libparakeet_amd_mel$w.so; [ $w = 16 ] && lib=parakeet.cpp_amd/libparakeet_amd.so
  rm -rf /tmp/pmcw$w
  PK_LIB=$PWD/$lib timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw$w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0 > /tmp/pmcw$w.log 2>&1
  python - $w >> $out <<PY
import csv, glob, sys
w = sys.argv[1]
tot, n = {}, {}
for f in glob.glob(f"/tmp/pmcw{w}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "WRITE_SIZE": continue
        k = r["Kernel_Name"].split("(")[0][:60]
        if "mel_" not in k: continue
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n.setdefault(k, set()).add(r["Dispatch_Id"])
for k in tot:
    print(f"waves={w} PMC WRITE_SIZE {k}: {tot[k] / len(n[k]) * 1024 / 1e6:.1f} MB per launch ({len(n[k])} launches; the log-mel tensor of 64 x 10 s is 20.5 MB)")
PY
done
cat $out
