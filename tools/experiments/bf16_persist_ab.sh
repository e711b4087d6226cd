#!/bin/bash
# tools/experiments/bf16_persist_ab.sh -- configs[2] (tdt-600m, 32 x 30 s, bf16): persistent direct-to-LDS GEMM (round 4) on / off, interleaved on one box
# (EXPERIMENTAL build: PK_BF16_PERSIST selects; the production library always runs the persistent form)
out=gpurun_out/bf16_persist_ab.txt
: > $out
for rep in 1 2 3; do
  for p in 0 1; do
    line=$(PK_LIB=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so PK_BF16_PERSIST=$p timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "persist=$p $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f frac=%.4f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'],r['frac'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','relpos_attention') if n in k))" "$line")" >> $out
  done
done
cat $out
