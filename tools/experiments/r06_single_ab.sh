#!/bin/bash
# tools/experiments/r06_single_ab.sh [LIB_A LIB_B] -- one 10 s clip end to end (tools/latency_single.py) and the decode kernels' average durations, library A against library B, interleaved
export TMPDIR=/tmp
o=gpurun_out/r06_single_ab; mkdir -p $o
A=${1:-$PWD/parakeet.cpp_amd/libparakeet_amd_prev.so}; B=${2:-$PWD/parakeet.cpp_amd/libparakeet_amd.so}
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_e2e.py tests/test_gpu_boost.py tests/test_gpu_ragged.py tests/test_gpu_stream.py tests/test_gpu_bf16.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
fi
: > $o/ab.txt
for rep in 1 2 3; do
  for l in A B; do
    lib=$A; [ $l = B ] && lib=$B
    echo "lib=$l $(PK_LIB=$lib timeout 200 python tools/latency_single.py 2>&1 | tr '\n' ' ')" >> $o/ab.txt
  done
done
for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  PK_LIB=$lib timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof$l -o kt -- python tools/latency_single.py > $o/prof$l.log 2>&1
  python tools/rocprof_summary.py $(ls $o/prof$l/*/kt_kernel_trace.csv $o/prof$l/kt_kernel_trace.csv 2>/dev/null | head -1) $o/kernel_stats_$l.md > /dev/null 2>&1
  rm -rf $o/prof$l
  echo "--- lib=$l" >> $o/ab.txt; grep "skinny\|decide" $o/kernel_stats_$l.md >> $o/ab.txt
done
cat $o/ab.txt; tail -3 $o/tests.log
