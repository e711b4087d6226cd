#!/bin/bash
# tools/experiments/lib_ab.sh -- A/B of two builds of the library on one box (PK_LIB selects the .so): default bench, interleaved
mkdir -p gpurun_out
out=gpurun_out/lib_ab.txt
: > $out
for rep in 1 2 3; do
  for lib in ${LIBS:-parakeet.cpp_amd/libparakeet_amd_prev.so parakeet.cpp_amd/libparakeet_amd.so}; do
    line=$(PK_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $BENCH_ARGS 2>/dev/null | tail -1)
    echo "$lib $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','sub_pw','sub_proj') if n in k))" "$line")" >> $out
  done
done
cat $out
