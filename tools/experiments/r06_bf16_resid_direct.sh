#!/bin/bash
# tools/experiments/r06_bf16_resid_direct.sh -- round 6 (round-5 verdict item 2): the bf16 residual products (fc2 / out_proj / pw2 of configs[2]) with the accumulators
# started from the residual and the epilogue from registers (gemm_bf16_glds.hpp: gl_resid_init / gl_epilogue_resid_direct).  Parity tests on the production
# library, then the interleaved A/B of the EXPERIMENTAL build: PK_BF16_FLAGS=2 (LDS epilogue, round 5) / 18 (register epilogue).
export TMPDIR=/tmp
o=gpurun_out/r06_bf16_resid; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_600m_depth.py tests/test_gpu_600m.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do
  for fl in 2 18; do
    line=$(PK_LIB=$exp PK_BF16_FLAGS=$fl timeout 300 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "flags=$fl $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','layernorm','relpos_attention') if n in k))" "$line")" >> $o/ab.txt
  done
done
cat $o/ab.txt; tail -5 $o/tests.log
