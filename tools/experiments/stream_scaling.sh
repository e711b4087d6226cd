#!/bin/bash
# tools/experiments/stream_scaling.sh TAG -- configs[4] at 1 .. 128 lock-step streams per GPU, exact fp32 mode and tolerance-class (bf16) mode:
# median latency per 160 ms chunk and aggregate RTFx.  Output: gpurun_out/TAG/stream_scaling.txt
o=gpurun_out/${1:-stream_scaling}
mkdir -p $o
: > $o/stream_scaling.txt
for n in ${STREAMS:-1 4 16 32 64 128}; do
  for mode in "" "--bf16"; do
    line=$(timeout 200 python tools/bench_stream.py --streams $n --chunks 60 --warmup 8 $mode 2>/dev/null | tail -1)
    echo "streams=$n mode=${mode:-fp32} $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('median_ms=%.3f p95_ms=%.3f rtfx=%.0f' % (d['latency_ms_median'], d['latency_ms_p95'], d['aggregate_rtfx']))" "$line" 2>/dev/null)" >> $o/stream_scaling.txt
  done
done
cat $o/stream_scaling.txt
