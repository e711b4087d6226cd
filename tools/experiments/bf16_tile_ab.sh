#!/bin/bash
# tools/experiments/bf16_tile_ab.sh -- tile variants of the direct-to-LDS bf16 GEMM (PK_BF16_TILE, EXPERIMENTAL builds) inside the engine:
# 0 = register-staged 128x128 kernel (round 2), 1 = 256x256 wide / 256x128 narrow, 2 = 256x128, 3 = 128x128 on 4 waves, 4 = 256x256 everywhere.
out=${1:-gpurun_out/bf16_tile_ab.txt}
: > "$out"
for m in 0 1 2 3 4 1 0; do
    line=$(PK_BF16_TILE=$m timeout 300 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | tail -1)
    echo "PK_BF16_TILE=$m | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["ms_per_step"], d["stage_ms"]["encoder"], {n:k[n]["ms"] for n in ("ffn_fc1_silu","ffn_fc2_resid","attn_qkv","attn_out_resid","conv_pw1_glu","conv_pw2_resid")}, d.get("parity",{}).get("per_clip"))' 2>/dev/null)" | tee -a "$out"
done
