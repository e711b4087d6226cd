#!/bin/bash
# tools/experiments/decode_group_ab.sh CONFIG-ARGS... -- decode group size / overlap A/B for one configuration (production build: API options only)
out=gpurun_out/decode_group_ab.txt
: > "$out"
for ov in 1 0; do for g in 1 4 8 16; do
    line=$(timeout 300 python bench.py "$@" --no-cpu-baseline --sustain-seconds 1.5 --decode-overlap $ov --decode-group $g 2>/dev/null | tail -1)
    echo "overlap=$ov group=$g | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["sustained"]["ms_per_step_median"], d["stage_ms"], {n:round(k[n]["ms"],2) for n in ("lstm_hh_cell","joint_pred_act","joint_heads_gemv","tdt_decide")})' 2>/dev/null)" | tee -a "$out"
done; done
