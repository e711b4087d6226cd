#!/bin/bash
# tools/experiments/r06_prewarm.sh -- round 6, verdict item 1: what would a streaming chunk's products cost if their weights had been requested for free?
# EXPERIMENTAL build, PK_SB_PREWARM = 0 (production chain) / 1 (a toucher launch in front of every product, same XCD -> L2-warm weights) / 2 (the toucher on
# another XCD -> memory-side cache only): rocprofv3 kernel trace of the streaming bench, per-kernel table of each.  The toucher's own time is not free; the
# PRODUCT kernels' durations under 1 / 2 against 0 are the ceiling of any cross-launch prefetch.
export TMPDIR=/tmp
o=gpurun_out/r06_prewarm; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 300 tools/ubench/l2_warm > $o/l2_warm_plain.txt 2>&1
for m in 0 1 2; do
  PK_LIB=$exp PK_SB_PREWARM=$m timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/p$m -o kt -- python tools/bench_stream.py --bf16 --chunks 45 --warmup 5 > $o/p$m.log 2>&1
  python tools/rocprof_summary.py $(ls $o/p$m/*/kt_kernel_trace.csv $o/p$m/kt_kernel_trace.csv 2>/dev/null | head -1) $o/stream_bf16_prewarm${m}_kernel_stats.md > /dev/null 2>&1
  rm -rf $o/p$m
  echo "prewarm=$m $(PK_LIB=$exp PK_SB_PREWARM=$m timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | cut -c150-330)" >> $o/chunk.txt
done
timeout 1200 python -m pytest tests -m gpu -q -x > $o/all_tests.log 2>&1
echo "all tests rc=$?" >> $o/all_tests.log
cat $o/l2_warm_plain.txt | head -30; cat $o/chunk.txt; for m in 0 1 2; do head -25 $o/stream_bf16_prewarm${m}_kernel_stats.md; done; tail -5 $o/all_tests.log
