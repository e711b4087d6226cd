#!/bin/bash
# tools/experiments/r06_pred_rows_bf16.sh -- the same predicate form in the bf16 decode products (decode_gemv_bf16.hip): bf16 / streaming parity tests, then the streaming
# chunk in both modes, library before (A: libparakeet_amd_prev2.so) against the current one (B), interleaved
export TMPDIR=/tmp
o=gpurun_out/r06_pred_bf16; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev2.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_stream.py tests/test_gpu_600m_depth.py tests/test_gpu_decode.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "stream-bf16 lib=$l $(PK_LIB=$lib timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['latency_ms_median'], d['latency_ms_p95'])")" >> $o/ab.txt
  echo "stream-fp32 lib=$l $(PK_LIB=$lib timeout 200 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['latency_ms_median'], d['latency_ms_p95'])")" >> $o/ab.txt
done; done
cat $o/ab.txt; tail -3 $o/tests.log
