#!/bin/bash
# tools/experiments/decode_modes_ab.sh -- headline configuration: decode off (CTC head only) / group sizes / loop forms, one line each
out=gpurun_out/decode_modes_ab.txt
: > "$out"
run() { line=$(timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 1.5 --steps 40 "$@" 2>/dev/null | tail -1); echo "$* | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["sustained"]["ms_per_step_median"])' 2>/dev/null)" | tee -a "$out"; }
run --decoder ctc
for g in 1 2 4 8 16; do run --decode-group $g; done
run --decode-group 4 --decode-overlap 0
run --decode-group 4 --decode-loop graph
run --decode-group 4 --decode-loop persistent
