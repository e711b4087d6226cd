#!/bin/bash
# tools/experiments/r05_bf16_blocked.sh -- the blocked fc1 -> fc2 activation hand-off (GemmArgs::out_blocked / a_blocked; active with
# PK_BF16_PERSIST=2, the production setting): phase stamps of fc1 with row-major and blocked stores, bit-equality of the encoder against the
# row-major LDS-epilogue form (PK_BF16_PERSIST=0), the bf16 parity tests, and the interleaved A/B of configs[2].
export TMPDIR=/tmp
o=gpurun_out/r05_bf16_blocked2
mkdir -p $o
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
./tools/ubench/gemm_bf16_trace 4096 1024 2 0 > $o/trace_rowmajor.txt 2>&1
./tools/ubench/gemm_bf16_trace 4096 1024 2 1 > $o/trace_blocked.txt 2>&1
grep -E "per launch|epilogue|whole tile|K loop" $o/trace_rowmajor.txt | head -5; grep -E "per launch|epilogue|whole tile|K loop" $o/trace_blocked.txt | head -5
PK_LIB=$exp PK_BF16_PERSIST=0 timeout 300 python tools/experiments/bf16_layout_check.py /tmp/enc_p0.npy > $o/layout_check.txt 2>&1
PK_LIB=$exp PK_BF16_PERSIST=2 timeout 300 python tools/experiments/bf16_layout_check.py /tmp/enc_p2.npy >> $o/layout_check.txt 2>&1
python -c "import numpy as np; a=np.load('/tmp/enc_p0.npy'); b=np.load('/tmp/enc_p2.npy'); print('encoder bits equal (row-major LDS epilogue vs blocked register epilogue):', bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), 'max |diff|', float(np.abs(a-b).max()))" >> $o/layout_check.txt 2>&1
tail -3 $o/layout_check.txt
PK_LIB=$exp PK_BF16_PERSIST=2 timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_600m_depth.py tests/test_gpu_ragged.py -m gpu -q -k "bf16" > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log; tail -2 $o/tests.log
out=$o/ab.txt
: > $out
for rep in 1 2 3; do
  for p in 0 3 2; do
    line=$(PK_LIB=$exp PK_BF16_PERSIST=$p timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "persist=$p $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f frac=%.4f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'],r['frac'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','relpos_attention') if n in k))" "$line")" >> $out
  done
done
cat $out
