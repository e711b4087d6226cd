#!/bin/bash
# round 5: a block's final LayerNorm folded into the next block's fc1 (streaming, tolerance-class mode): unit tests, A/B, streaming parity, kernel table.
o=gpurun_out/r05_fuse_final; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "two_folded or folded_layernorm" 2>&1 | tail -5 > $o/unit.txt; cat $o/unit.txt
ab() {
    echo "fuse_fin=$1  $(PK_LIB=$exp PK_STREAM_FUSE_FIN=$1 timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("latency_ms_median","latency_ms_p95","aggregate_rtfx","weight_stream_tbps") if k in d})')" >> $o/ab.txt
}
: > $o/ab.txt
for r in 1 2; do ab 0; ab 1; done
cat $o/ab.txt
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -5 > $o/tests.txt; cat $o/tests.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python tools/bench_stream.py --bf16 --chunks 45 --warmup 5 > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/kernel_stats.md > /dev/null 2>&1
head -12 $o/kernel_stats.md
rm -rf $o/prof
