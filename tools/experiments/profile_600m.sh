#!/bin/bash
# tools/experiments/profile_600m.sh TAG -- configs[2] bench line + rocprofv3 --kernel-trace --stats summary of the same command (one gpurun call)
o=gpurun_out/$1
mkdir -p $o
export TMPDIR=/tmp
timeout 400 python bench.py --config tdt-600m --bf16 > $o/bench_600m_bf16.json 2> $o/bench_600m_bf16.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof600 -o kt -- python bench.py --config tdt-600m --bf16 --steps 4 --warmup 1 --no-cpu-baseline --sustain-seconds 0 > $o/prof600.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof600/*/kt_kernel_trace.csv $o/prof600/kt_kernel_trace.csv 2>/dev/null | head -1) $o/kernel_stats_600m_bf16.md > /dev/null 2>&1
rm -rf $o/prof600
head -c 400 $o/bench_600m_bf16.json; echo; head -24 $o/kernel_stats_600m_bf16.md
