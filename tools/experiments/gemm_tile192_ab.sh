#!/bin/bash
# tools/experiments/gemm_tile192_ab.sh -- 192x128 tiles for products whose 128x128 tile count falls between one and two rounds of the 512 resident
# workgroups (PK_GEMM_VARIANT bit 128, kernels/gemm.hip): parity under the variant, then the default bench per mask, interleaved on one box.
mkdir -p gpurun_out
out=gpurun_out/gemm_tile192_ab.txt
: > $out
export PK_LIB=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
PK_GEMM_VARIANT=203 timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_ragged.py tests/test_gpu_encoder.py -m gpu -x -q 2>&1 | tail -3 >> $out
for rep in 1 2 3; do
  for m in 75 203; do
    line=$(PK_GEMM_VARIANT=$m timeout 120 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 3 2>/dev/null | tail -1)
    echo "mask=$m $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f | ms/step: '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','sub_pw','sub_proj') if n in k))" "$line")" >> $out
  done
done
cat $out
