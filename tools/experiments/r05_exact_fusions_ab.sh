#!/bin/bash
# round 5, after the register fix of the exact chain kernel: do the two fusions still pay in the EXACT mode at 16 sessions?  (EXPERIMENTAL build switches)
o=gpurun_out/r05_exact_fusions; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/ab.txt
for rep in 1 2 3; do
  for sw in "0 0" "1 0" "0 1" "1 1"; do
    set -- $sw
    line=$(PK_LIB=$exp PK_STREAM_FUSE_DW=$1 PK_STREAM_FUSE_FIN=$2 timeout 200 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1)
    echo "fp32 16 sessions fuse_dw=$1 fuse_fin=$2 $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('median_ms=%.3f p95=%.3f' % (d['latency_ms_median'], d['latency_ms_p95']))" "$line" 2>/dev/null)" >> $o/ab.txt
  done
done
cat $o/ab.txt
