#!/bin/bash
# tools/experiments/r05_bf16_ring.sh -- the continuous-stream kernel (gemm_bf16_ring.hpp, PK_BF16_PERSIST=4) against =0 / =2; as r05_bf16_direct.sh: configs[2] (tdt-600m, 32 x 30 s, bf16): the DIRECT register epilogue of the direct-to-LDS GEMM
# (gemm_bf16_glds.hpp, round 5), persistent (PK_BF16_PERSIST=2) and one tile per workgroup (=3), against the LDS epilogue (=0): first the
# parity tests of the bf16 mode under each switch, then the interleaved A/B; then the vendor yardstick on this box (rocBLAS, same shapes).
export TMPDIR=/tmp
o=gpurun_out/r05_bf16_ring
mkdir -p $o
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
for p in 4; do
  PK_LIB=$exp PK_BF16_PERSIST=$p timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_600m_depth.py -m gpu -q -k "bf16" > $o/tests_p$p.log 2>&1
  echo "persist=$p tests rc=$?" >> $o/tests_p$p.log
  tail -2 $o/tests_p$p.log
done
out=$o/ab.txt
: > $out
for rep in 1 2 3; do
  for p in 0 2 4; do
    line=$(PK_LIB=$exp PK_BF16_PERSIST=$p timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "persist=$p $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f frac=%.4f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'],r['frac'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','relpos_attention') if n in k))" "$line")" >> $out
  done
done
cat $out
