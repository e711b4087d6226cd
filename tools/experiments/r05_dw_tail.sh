#!/bin/bash
# round 5: depthwise conv + BatchNorm + SiLU in the GLU epilogue of pw1 (streaming, tolerance-class mode): unit test, A/B, stream parity tests.
#    gpurun -- bash tools/experiments/r05_dw_tail.sh
o=gpurun_out/r05_dw_tail; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "depthwise or folded" 2>&1 | tail -5 > $o/unit.txt; cat $o/unit.txt
ab() {
    echo "fuse_dw=$1  $(PK_LIB=$exp PK_STREAM_FUSE_DW=$1 timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("latency_ms_median","latency_ms_p95","aggregate_rtfx","weight_stream_tbps") if k in d})')" >> $o/ab.txt
}
: > $o/ab.txt
for r in 1 2; do ab 0; ab 1; done
cat $o/ab.txt
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -5 > $o/tests.txt; cat $o/tests.txt
