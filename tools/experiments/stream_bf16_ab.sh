#!/bin/bash
# tools/experiments/stream_bf16_ab.sh TAG -- configs[4] (nemotron-600m, 16 lock-step streams, 160 ms chunks): the exact fp32 mode, the tolerance-class
# mode (bf16 operands: kernels/gemm_smallm_bf16.hip) and, in the EXPERIMENTAL build, that mode with the folded LayerNorm switched off
# (PK_STREAM_FUSE_LN=0), interleaved on one box; then rocprofv3 --kernel-trace --stats of the bf16 bench.  Output: gpurun_out/TAG/
o=gpurun_out/${1:-stream_bf16_ab}
mkdir -p $o
export TMPDIR=/tmp
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/ab.txt
for rep in 1 2; do
  echo "fp32          $(timeout 200 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1 | cut -c1-420)" >> $o/ab.txt
  echo "bf16 fused-LN $(PK_LIB=$exp PK_STREAM_FUSE_LN=1 timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | cut -c1-420)" >> $o/ab.txt
  echo "bf16 plain-LN $(PK_LIB=$exp PK_STREAM_FUSE_LN=0 timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | cut -c1-420)" >> $o/ab.txt
done
echo "bf16 64 streams $(timeout 200 python tools/bench_stream.py --bf16 --streams 64 --chunks 60 --warmup 10 2>/dev/null | tail -1 | cut -c1-420)" >> $o/ab.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python tools/bench_stream.py --bf16 --chunks 45 --warmup 5 > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/stream_bf16_kernel_stats.md > /dev/null 2>&1
rm -rf $o/prof
cat $o/ab.txt; head -24 $o/stream_bf16_kernel_stats.md
