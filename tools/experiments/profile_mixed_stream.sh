export TMPDIR=/tmp
o=gpurun_out/r04_extra
mkdir -p $o
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_mixed -o kt -- python tools/bench_mixed.py --steps 5 --warmup 1 --clips 64 --oracle-sample 0 > $o/mixed.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof_mixed/*/kt_kernel_trace.csv $o/prof_mixed/kt_kernel_trace.csv 2>/dev/null | head -1) $o/mixed_kernel_stats.md > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_stream -o kt -- python tools/bench_stream.py --chunks 50 --warmup 5 > $o/stream.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof_stream/*/kt_kernel_trace.csv $o/prof_stream/kt_kernel_trace.csv 2>/dev/null | head -1) $o/stream_kernel_stats.md > /dev/null 2>&1
rm -rf $o/prof_mixed $o/prof_stream
tail -2 $o/mixed.log | cut -c1-400; tail -1 $o/stream.log; head -16 $o/mixed_kernel_stats.md; head -14 $o/stream_kernel_stats.md
