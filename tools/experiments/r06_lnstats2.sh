#!/bin/bash
# tools/experiments/r06_lnstats2.sh -- second form of the LayerNorm-from-statistics fold (normalisation moved out of the barrier-to-barrier section, packed VALU):
# interleaved A/B, EXPERIMENTAL builds: exp (normalise under the LAST sub-step before the barrier) / exp2 (-DGP_LNORM_AT=0: under the first), PK_LN_STATS=0 = separate launches.
export TMPDIR=/tmp
o=gpurun_out/r06_lnstats2; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "statistics" > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do
  for cfg in "exp 0" "exp 1" "exp2 1"; do
    set -- $cfg
    line=$(PK_LIB=$PWD/parakeet.cpp_amd/libparakeet_amd_$1.so PK_LN_STATS=$2 timeout 200 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "lib=$1 ln_stats=$2 $(python -c "import json,sys; d=json.loads(sys.argv[1]); r=d['roofline']; k=d['kernels']; print('ms_per_step=%.3f enc=%.3f fc1_us=%.1f | '%(d['ms_per_step'],d['stage_ms']['encoder'],r['us_per_launch'])+' '.join('%s=%.3f'%(n.replace('ffn_','').replace('attn_','').replace('conv_',''),k[n]['ms']) for n in ('ffn_fc1_silu','ffn_fc2_resid','attn_qkv','conv_pw1_glu','layernorm','layernorm_stats','layernorm_then_stats') if n in k))" "$line")" >> $o/ab.txt
  done
done
cat $o/ab.txt; tail -3 $o/tests.log
