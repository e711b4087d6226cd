#!/bin/bash
# round 5: the conv tail in the exact mode's GLU kernel (gemm_smallm_ln_kernel): A/B and the streaming parity tests (fp32 rows bit-identical to the oracle).
o=gpurun_out/r05_dw_tail; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
ab() {
    echo "fp32 fuse_dw=$1  $(PK_LIB=$exp PK_STREAM_FUSE_DW=$1 timeout 200 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("latency_ms_median","latency_ms_p95","aggregate_rtfx","weight_stream_tbps") if k in d})')" >> $o/ab_fp32.txt
}
: > $o/ab_fp32.txt
for r in 1 2; do ab 0; ab 1; done
cat $o/ab_fp32.txt
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -5 > $o/tests_fp32.txt; cat $o/tests_fp32.txt
