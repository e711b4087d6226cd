#!/bin/bash
# tools/experiments/r06_lnstats.sh -- round 6: LayerNorm of large fp32 batches as a statistics pass + normalise-on-stage in the tile GEMM (gemm_pipe.hpp LNA).
# Parity tests, then the interleaved A/B on one box (EXPERIMENTAL build: PK_LN_STATS=0 keeps the separate LayerNorm launches).
export TMPDIR=/tmp
o=gpurun_out/r06_lnstats; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_encoder.py tests/test_gpu_e2e.py tests/test_gpu_ragged.py tests/test_gpu_600m.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do
  for sw in 0 1; do
    line=$(PK_LIB=$exp PK_LN_STATS=$sw timeout 200 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "ln_stats=$sw $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','attn_out_resid','conv_pw1_glu','conv_pw2_resid','layernorm','layernorm_stats','layernorm_then_stats','relpos_attention') if n in k))" "$line")" >> $o/ab.txt
  done
done
cat $o/ab.txt; tail -5 $o/tests.log
