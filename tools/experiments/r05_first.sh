#!/bin/bash
# tools/experiments/r05_first.sh -- round 5, first GPU call: the new parity tests with their printed figures, the whole GPU suite, the default
# bench line, and the small-M bf16 column-tile A/B (EXPERIMENTAL build: PK_SB_CT / PK_SB_NT) on the streaming bench.
export TMPDIR=/tmp
o=gpurun_out/r05_first
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_600m_depth.py tests/test_gpu_stream.py tests/test_gpu_ragged.py tests/test_gpu_group.py -m gpu -q -s -k "teacher or fp32_reference or score or bf16_mode_vs_bf16_oracle or long_file or two_rank or two_ranks" > $o/new_tests.log 2>&1
echo "new tests rc=$?" >> $o/new_tests.log
timeout 900 python -m pytest tests -m gpu -q > $o/all_tests.log 2>&1
echo "all tests rc=$?" >> $o/all_tests.log
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/sb_ab.txt
for rep in 1 2; do
  for cfgs in "0 0" "2 0" "0 1" "2 1"; do
    set -- $cfgs
    echo "ct=$1 nt=$2 $(PK_LIB=$exp PK_SB_CT=$1 PK_SB_NT=$2 timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | cut -c150-330)" >> $o/sb_ab.txt
  done
done
PK_LIB=$exp PK_SB_CT=2 PK_SB_NT=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python tools/bench_stream.py --bf16 --chunks 45 --warmup 5 > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/stream_bf16_ct2_kernel_stats.md > /dev/null 2>&1
rm -rf $o/prof
timeout 500 python bench.py > $o/bench.json 2> $o/bench.err
tail -5 $o/new_tests.log; tail -3 $o/all_tests.log; cat $o/sb_ab.txt; head -c 400 $o/bench.json
