#!/bin/bash
# tools/experiments/r06_single_profile.sh -- rocprofv3 --kernel-trace --stats of Transcriber::transcribe on ONE 10 s clip (tools/latency_single.py: 21 TDT + 21 CTC calls)
export TMPDIR=/tmp
o=gpurun_out/r06_single; mkdir -p $o
timeout 300 python tools/latency_single.py > $o/latency.txt 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python tools/latency_single.py > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/single_kernel_stats.md > /dev/null 2>&1
cp $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/kt.csv; gzip -f $o/kt.csv
rm -rf $o/prof
cat $o/latency.txt; head -45 $o/single_kernel_stats.md
