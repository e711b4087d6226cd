#!/bin/bash
# tools/experiments/r05_bf16_resid_init.sh -- bf16 residual products accumulated ONTO the residual (GemmArgs::resid_init; EXPERIMENTAL build:
# PK_BF16_FLAGS=10 on / 2 off): unit tests, interleaved A/B of configs[2], the depth-24 parity tests.
export TMPDIR=/tmp
o=gpurun_out/r05_bf16_resid_init
mkdir -p $o
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "glds" 2>&1 | tail -4 > $o/unit.txt; cat $o/unit.txt
out=$o/ab.txt
: > $out
for rep in 1 2 3; do
  for f in 2 10; do
    line=$(PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=$f timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "flags=$f $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','relpos_attention') if n in k))" "$line")" >> $out
  done
done
cat $out
timeout 900 python -m pytest tests/test_gpu_600m_depth.py tests/test_gpu_ragged.py tests/test_gpu_bf16.py -m gpu -x -q -k "bf16" 2>&1 | tail -4 > $o/tests.txt; cat $o/tests.txt
