#!/bin/bash
# tools/experiments/r06_wave_xor.sh -- wave reductions on DPP / permlane exchanges (pk_devmath.h wave_xor) + the decode window: the whole GPU suite on the production
# library, then previous library (A) against this one (B), interleaved: headline step, streaming chunk (both modes), one clip end to end; window sizes in the EXPERIMENTAL build
export TMPDIR=/tmp
o=gpurun_out/r06_wave_xor; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
fi
: > $o/ab.txt
for rep in 1 2 3; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  line=$(PK_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "headline lib=$l $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f enc=%.3f attention=%.3f layernorm=%.3f decide=%.3f' % (d['ms_per_step'], d['stage_ms']['encoder'], k['relpos_attention']['ms'], k['layernorm']['ms'], k['tdt_decide']['ms']))" "$line")" >> $o/ab.txt
done; done
for rep in 1 2; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "stream-bf16 lib=$l $(PK_LIB=$lib timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['latency_ms_median'], d['latency_ms_p95'])")" >> $o/ab.txt
  echo "stream-fp32 lib=$l $(PK_LIB=$lib timeout 200 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['latency_ms_median'], d['latency_ms_p95'])")" >> $o/ab.txt
done; done
for rep in 1 2 3; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "single lib=$l $(PK_LIB=$lib timeout 200 python tools/latency_single.py 2>&1 | tr '\n' ' ')" >> $o/ab.txt
done; done
for rep in 1 2; do for sw in 0 2 4; do for b in 1 2 4; do
  line=$(PK_LIB=$exp PK_DEC_WIN=$sw timeout 300 python bench.py --batch $b --no-cpu-baseline --no-also --steps 50 --warmup 5 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "batch=$b dec_win=$sw $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('ms_per_step=%.3f stage_ms=%s' % (d['ms_per_step'], d['stage_ms']))" "$line")" >> $o/ab.txt
done; done; done
cat $o/ab.txt; tail -5 $o/tests.log
