#!/bin/bash
# round 5: the small-M bf16 kernel on operand-tiled weights (GemmArgs::W_t16) and with two column tiles per wave for the wide fp32-row products
# (fc1 / qkv): A/B through the switches of the EXPERIMENTAL build, then the streaming bf16 parity tests on the production library, then a
# kernel table.    gpurun -- bash tools/experiments/r05_sb_tiles.sh
o=gpurun_out/r05_sb_tiles; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
ab() {  # wt ct
    echo "wt=$1 ct=$2  $(PK_LIB=$exp PK_SB_WT=$1 PK_SB_CT=$2 timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("latency_ms_median","latency_ms_p95","aggregate_rtfx","weight_stream_tbps") if k in d})')" >> $o/ab.txt
}
: > $o/ab.txt
for r in 1 2; do ab 0 1; ab 1 1; ab 0 0; ab 1 0; done
cat $o/ab.txt
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -5 > $o/tests.txt; cat $o/tests.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python tools/bench_stream.py --bf16 --chunks 45 --warmup 5 > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/kernel_stats.md > /dev/null 2>&1
head -20 $o/kernel_stats.md
rm -rf $o/prof
