#!/bin/bash
# tools/experiments/r06_decide_fast.sh -- round 6 (round-5 verdict item 6): the tolerance-class mode's greedy decision on the register-resident form (decode_dev.hpp FAST).
# bf16-mode parity tests on the production library, then the interleaved A/B of the EXPERIMENTAL build (PK_DEC_FAST=0 / 1) on configs[2] and the streaming bench.
export TMPDIR=/tmp
o=gpurun_out/r06_decide; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_600m_depth.py tests/test_gpu_stream.py tests/test_gpu_ragged.py tests/test_gpu_decode.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do
  for sw in 0 1; do
    line=$(PK_LIB=$exp PK_DEC_FAST=$sw timeout 300 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "600m-bf16 dec_fast=$sw $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f stage_ms=%s tdt_decide=%.3f' % (d['ms_per_step'], d['stage_ms'], k.get('tdt_decide',{}).get('ms',0)))" "$line")" >> $o/ab.txt
  done
done
for sw in 0 1; do echo "stream-bf16 dec_fast=$sw $(PK_LIB=$exp PK_DEC_FAST=$sw timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | cut -c150-260)" >> $o/ab.txt; done
cat $o/ab.txt; tail -4 $o/tests.log
