#!/bin/bash
# tools/experiments/r06_b1_stage.sh -- batch of ONE 10 s clip through bench.py: stage times and event-timed kernel table, library A / B and the decode-loop modes
export TMPDIR=/tmp
o=gpurun_out/r06_b1; mkdir -p $o
A=${1:-$PWD/parakeet.cpp_amd/libparakeet_amd_prev.so}; B=${2:-$PWD/parakeet.cpp_amd/libparakeet_amd.so}
: > $o/b1.txt
for rep in 1 2; do for l in A B; do for lp in phases graph; do
  lib=$A; [ $l = B ] && lib=$B
  line=$(PK_LIB=$lib timeout 300 python bench.py --batch 1 --decode-loop $lp --no-cpu-baseline --no-also --steps 50 --warmup 5 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "lib=$l loop=$lp $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f stage_ms=%s' % (d['ms_per_step'], d['stage_ms']), ' '.join('%s=%.3f' % (n, k[n]['ms']) for n in ('lstm_hh_cell','joint_pred_act','joint_heads_gemv','tdt_decide') if n in k))" "$line")" >> $o/b1.txt
done; done; done
cat $o/b1.txt
