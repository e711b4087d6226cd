#!/bin/bash
# tools/experiments/profile_single.sh TAG [SECONDS] -- rocprofv3 --kernel-trace --stats of the reference's protocol (encoder only, batch 1) on tdt-ctc-110m
o=gpurun_out/$1
mkdir -p $o
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python tools/bench_reference_protocol.py --models tdt-ctc-110m --seconds ${2:-10} --iters 20 > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/single_kernel_stats.md > /dev/null 2>&1
rm -rf $o/prof
head -30 $o/single_kernel_stats.md
