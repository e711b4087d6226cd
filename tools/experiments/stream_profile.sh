#!/bin/bash
# tools/experiments/stream_profile.sh TAG -- streaming bench line + rocprofv3 --kernel-trace --stats summary of the same command (one gpurun call)
o=gpurun_out/$1
mkdir -p $o
export TMPDIR=/tmp
timeout 300 python tools/bench_stream.py 2>/dev/null | tail -1 > $o/stream.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python tools/bench_stream.py --chunks 45 --warmup 5 > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/stream_kernel_stats.md > /dev/null 2>&1
rm -rf $o/prof
cat $o/stream.json; head -16 $o/stream_kernel_stats.md
