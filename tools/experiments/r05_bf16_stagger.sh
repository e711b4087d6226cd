#!/bin/bash
# tools/experiments/r05_bf16_stagger.sh -- configs[2]: wave-role flags of the direct-to-LDS bf16 GEMMs (PK_BF16_FLAGS=1: the two waves of a SIMD
# request their DMA pieces at different points of the K tile; a template instantiation, second pass after the run-time-flag build) on the persistent form (PK_BF16_PERSIST=2)
# and the continuous-stream form (=4): parity tests under the flags, then the interleaved A/B.
export TMPDIR=/tmp
o=gpurun_out/r05_bf16_stagger2
mkdir -p $o
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=1 timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_600m_depth.py -m gpu -q -k "bf16" > $o/tests_p2f1.log 2>&1; echo "p2 f1 rc=$?" >> $o/tests_p2f1.log; tail -2 $o/tests_p2f1.log
PK_LIB=$exp PK_BF16_PERSIST=4 PK_BF16_FLAGS=1 timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_600m_depth.py -m gpu -q -k "bf16" > $o/tests_p4f1.log 2>&1; echo "p4 f1 rc=$?" >> $o/tests_p4f1.log; tail -2 $o/tests_p4f1.log
out=$o/ab.txt
: > $out
for rep in 1 2 3; do
  for cfgs in "2 0" "2 1" "4 0" "4 1"; do
    set -- $cfgs
    line=$(PK_LIB=$exp PK_BF16_PERSIST=$1 PK_BF16_FLAGS=$2 timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "persist=$1 flags=$2 $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f frac=%.4f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'],r['frac'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','relpos_attention') if n in k))" "$line")" >> $out
  done
done
cat $out
