#!/bin/bash
# tools/experiments/decode_overlap_ab.sh -- how should the TDT loop be scheduled against the encoder?  A/B of pk_batch_set_decode_overlap,
# the decode group size and (EXPERIMENTAL builds: PK_DEC_NT) non-temporal weight loads in the decode GEMVs, for both benchmark configurations.
# Writes one line per run: config, knobs, ms/step (timed), sustained median.   usage (GPU box): bash tools/experiments/decode_overlap_ab.sh OUT
out=${1:-gpurun_out/decode_overlap_ab.txt}
run() {  # label, env, args...
    label=$1; envs=$2; shift 2
    line=$(env $envs timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 1.5 "$@" 2>/dev/null | tail -1)
    echo "$label | $envs | $* | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["sustained"]["ms_per_step_median"], d["stage_ms"]["encoder"])' 2>/dev/null)" | tee -a "$out"
}
: > "$out"
for cfg in "--config tdt-600m --bf16" ""; do
    run base "A=0" $cfg --decode-overlap 1 --decode-group 4
    run serial "A=0" $cfg --decode-overlap 0 --decode-group 4
    run serial8 "A=0" $cfg --decode-overlap 0 --decode-group 8
    run group8 "A=0" $cfg --decode-overlap 1 --decode-group 8
    run group16 "A=0" $cfg --decode-overlap 1 --decode-group 16
    run nt "PK_DEC_NT=1" $cfg --decode-overlap 1 --decode-group 4
    run nt8 "PK_DEC_NT=1" $cfg --decode-overlap 1 --decode-group 8
done
