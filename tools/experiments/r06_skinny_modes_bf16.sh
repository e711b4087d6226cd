#!/bin/bash
# tools/experiments/r06_skinny_modes_bf16.sh -- launch modes as template arguments in the bf16 decode products: bf16 parity tests, then configs[2] and the bf16 streaming chunk,
# library before (A: libparakeet_amd_prev2.so) / after (B), interleaved
export TMPDIR=/tmp
o=gpurun_out/r06_modes_bf16; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev2.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_stream.py tests/test_gpu_600m_depth.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  line=$(PK_LIB=$lib timeout 300 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "600m-bf16 lib=$l $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('ms_per_step=%.3f decode_stage=%.3f' % (d['ms_per_step'], d['stage_ms']['decode']))" "$line")" >> $o/ab.txt
  echo "stream-bf16 lib=$l $(PK_LIB=$lib timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['latency_ms_median'], d['latency_ms_p95'])")" >> $o/ab.txt
done; done
cat $o/ab.txt; tail -3 $o/tests.log
