#!/bin/bash
# tools/experiments/r06_first.sh -- round 6, first GPU call: (1) does a weight tile requested by an EARLIER launch arrive faster (tools/ubench/l2_warm),
# (2) the streaming bench on 24- / 4- / 2-block cuts (a 4-block cut's bf16 weights, 193 MB, stay in the memory-side cache from chunk to chunk: what
# does a block cost when its weights are not cold?), (3) the default bench line of this box.
export TMPDIR=/tmp
o=gpurun_out/r06_first
mkdir -p $o
timeout 300 tools/ubench/l2_warm > $o/l2_warm.txt 2>&1
: > $o/stream_layers.txt
for L in 24 4 2; do
  for f in "--bf16" ""; do
    echo "layers=$L $f $(timeout 300 python tools/bench_stream.py $f --layers $L --chunks 150 --warmup 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["latency_ms_median"], d["latency_ms_p95"], d["latency_ms_mean"])')" >> $o/stream_layers.txt
  done
done
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err
cat $o/l2_warm.txt; cat $o/stream_layers.txt; head -c 600 $o/bench.json
