#!/bin/bash
# tools/experiments/r05_bf16_asmfrag.sh -- hand-counted fragment reads in the direct-to-LDS bf16 GEMMs (gemm_bf16_glds.hpp ASMFRAG; EXPERIMENTAL
# build: PK_BF16_FLAGS=2): encoder bits against the compiler-scheduled form, the bf16 parity tests, the interleaved A/B of configs[2].
export TMPDIR=/tmp
o=gpurun_out/r05_bf16_asmfrag
mkdir -p $o
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
PK_LIB=$exp PK_BF16_PERSIST=0 PK_BF16_FLAGS=0 timeout 300 python tools/experiments/bf16_layout_check.py /tmp/enc_a.npy > $o/bits.txt 2>&1
PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=2 timeout 300 python tools/experiments/bf16_layout_check.py /tmp/enc_b.npy >> $o/bits.txt 2>&1
PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=2 timeout 300 python tools/experiments/bf16_layout_check.py /tmp/enc_c.npy >> $o/bits.txt 2>&1
python -c "import numpy as np; a=np.load('/tmp/enc_a.npy'); b=np.load('/tmp/enc_b.npy'); c=np.load('/tmp/enc_c.npy'); print('encoder bits equal (compiler-scheduled LDS epilogue vs ASMFRAG persistent, two runs):', bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), bool(np.array_equal(a.view(np.uint32), c.view(np.uint32))))" >> $o/bits.txt 2>&1
tail -2 $o/bits.txt
PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=2 timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_600m_depth.py tests/test_gpu_ragged.py -m gpu -q -k "bf16" > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log; tail -2 $o/tests.log
out=$o/ab.txt
: > $out
for rep in 1 2 3; do
  for f in 0 2; do
    line=$(PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=$f timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "flags=$f $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f frac=%.4f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'],r['frac'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','relpos_attention') if n in k))" "$line")" >> $out
  done
done
cat $out
