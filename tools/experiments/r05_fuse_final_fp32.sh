#!/bin/bash
# round 5: the final LayerNorm folded into the next block's fc1 in the EXACT mode (gemm_smallm_ln_kernel): A/B and the streaming parity tests (bit-identical fixtures).
o=gpurun_out/r05_fuse_final; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
ab() {
    echo "fp32 fuse_fin=$1  $(PK_LIB=$exp PK_STREAM_FUSE_FIN=$1 timeout 200 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("latency_ms_median","latency_ms_p95","aggregate_rtfx","weight_stream_tbps") if k in d})')" >> $o/ab_fp32.txt
}
: > $o/ab_fp32.txt
for r in 1 2; do ab 0; ab 1; done
cat $o/ab_fp32.txt
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_sortformer.py -m gpu -x -q 2>&1 | tail -5 > $o/tests_fp32.txt; cat $o/tests_fp32.txt
