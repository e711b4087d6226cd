#!/bin/bash
# round 5: which of the streaming fusions (conv tail in pw1's epilogue, final LayerNorm folded into the next block's fc1) pay at 32 / 64 / 128 sessions?
# EXPERIMENTAL build switches, both modes.
o=gpurun_out/r05_stream_fusions; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/ab.txt
for n in ${STREAMS:-16 32 64 128}; do
  for mode in "" "--bf16"; do
    for sw in "1 1" "0 1" "1 0" "0 0"; do
      set -- $sw
      line=$(PK_LIB=$exp PK_STREAM_FUSE_DW=$1 PK_STREAM_FUSE_FIN=$2 timeout 200 python tools/bench_stream.py --streams $n --chunks 50 --warmup 8 $mode 2>/dev/null | tail -1)
      echo "streams=$n mode=${mode:-fp32} fuse_dw=$1 fuse_fin=$2 $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('median_ms=%.3f rtfx=%.0f' % (d['latency_ms_median'], d['aggregate_rtfx']))" "$line" 2>/dev/null)" >> $o/ab.txt
    done
  done
done
cat $o/ab.txt
