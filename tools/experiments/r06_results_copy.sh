#!/bin/bash
# tools/experiments/r06_results_copy.sh -- results leave through one pinned staging area on a stream of their own (capi.cpp copy_results): tests, then library
# before (A: libparakeet_amd_prev2.so) against after (B): pk_transcribe_pcm of 1 / 8 clips per call (median of 100 calls), the PCIe-inclusive headline line
export TMPDIR=/tmp
o=gpurun_out/r06_rescopy; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev2.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so
timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ragged.py tests/test_gpu_facade.py tests/test_gpu_group.py tests/test_gpu_vs_reference_code.py tests/test_gpu_decode.py tests/test_gpu_boost.py tests/test_gpu_bench_launcher.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do for n in 1 8; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "clips=$n lib=$l $(PK_LIB=$lib PK_LAT_CLIPS=$n timeout 200 python tools/latency_single.py 2>&1 | tr '\n' ' ')" >> $o/ab.txt
done; done; done
cat $o/ab.txt; tail -3 $o/tests.log
