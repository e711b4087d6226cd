#!/bin/bash
# tools/experiments/r06_pred_rows.sh -- need flags as predicates in the cell / joint-activation products of small lock-step batches (decode_dev.hpp: PRED): decode tests, then
# the library before (A: libparakeet_amd_prev2.so) against the current one (B), pk_transcribe_pcm of 1 / 2 / 4 / 8 clips per call, median of 100 calls, interleaved
export TMPDIR=/tmp
o=gpurun_out/r06_pred; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev2.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_e2e.py tests/test_gpu_ragged.py tests/test_gpu_stream.py tests/test_gpu_boost.py tests/test_gpu_vs_reference_code.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do for n in 1 2 4 8; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "clips=$n lib=$l $(PK_LIB=$lib PK_LAT_CLIPS=$n timeout 200 python tools/latency_single.py 2>&1 | head -1)" >> $o/ab.txt
done; done; done
cat $o/ab.txt; tail -3 $o/tests.log
