#!/bin/bash
# tools/experiments/r05_bf16_rowblock.sh -- the persistent bf16 GEMM walking a block of tile ROWS per XCD (gemm_bf16_glds.hpp rowblock; EXPERIMENTAL
# build: PK_BF16_FLAGS=6 on / 2 off): encoder bits, interleaved A/B of configs[2], FETCH_SIZE / WRITE_SIZE per launch of fc1 for both.
export TMPDIR=/tmp
o=gpurun_out/r05_bf16_rowblock
mkdir -p $o
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=2 timeout 300 python tools/experiments/bf16_layout_check.py /tmp/enc_a.npy > $o/bits.txt 2>&1
PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=6 timeout 300 python tools/experiments/bf16_layout_check.py /tmp/enc_b.npy >> $o/bits.txt 2>&1
python -c "import numpy as np; a=np.load('/tmp/enc_a.npy'); b=np.load('/tmp/enc_b.npy'); print('encoder bits equal (contiguous tile range per XCD vs row block per XCD):', bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))))" >> $o/bits.txt 2>&1
tail -1 $o/bits.txt
out=$o/ab.txt
: > $out
for rep in 1 2 3; do
  for f in 2 6; do
    line=$(PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=$f timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "flags=$f $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f frac=%.4f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'],r['frac'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','relpos_attention') if n in k))" "$line")" >> $out
  done
done
cat $out
B="--config tdt-600m --bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0"
for f in 2 6; do
  PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=$f timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/f$f -o f -- python bench.py $B > $o/f$f.log 2>&1
  PK_LIB=$exp PK_BF16_PERSIST=2 PK_BF16_FLAGS=$f timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/w$f -o w -- python bench.py $B > $o/w$f.log 2>&1
  python tools/pmc_hbm.py $o/f$f $o/w$f $o/pmc_hbm_flags$f.json "gemm_bf16_glds_kernel<4, 2, 2, 4, 2" 12032 4096 2 > /dev/null 2>&1
  rm -rf $o/f$f $o/w$f $o/f$f.log $o/w$f.log
  python -c "import json; j=json.load(open('$o/pmc_hbm_flags$f.json')); print('flags=$f', {k: j[k] for k in j if k != 'kernels'}); [print('   ', k, v) for k, v in j['kernels'].items() if 'glds' in k]"
done
