#!/bin/bash
# tools/experiments/gemm_variant_ab.sh -- A/B of GEMM tile variants inside the engine (PK_GEMM_VARIANT bits, kernels/gemm.hip):
# the default bench on one box, once per mask, interleaved twice so that box drift shows.  Output: gpurun_out/gemm_variant_ab.txt
mkdir -p gpurun_out
out=gpurun_out/gemm_variant_ab.txt
: > $out
for rep in 1 2; do
  for m in ${MASKS:-0 1 2 4 8 15 0}; do
    line=$(PK_GEMM_VARIANT=$m timeout 120 python bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | tail -1)
    echo "mask=$m $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f | ms/step: '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','sub_pw','sub_proj') if n in k))" "$line")" >> $out
  done
done
cat $out
