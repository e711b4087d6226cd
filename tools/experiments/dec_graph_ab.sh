#!/bin/bash
# tools/experiments/dec_graph_ab.sh -- decode loop as per-phase launches vs a replayed hipGraph of 16 steps (PK_DEC_GRAPH), default bench.
mkdir -p gpurun_out
out=gpurun_out/dec_graph_ab.txt
: > $out
for rep in 1 2; do
  for g in 0 1; do
    line=$(PK_DEC_GRAPH=$g timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1)
    echo "graph=$g $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('ms_per_step=%.3f rtfx=%.0f stage_ms=%s'%(d['ms_per_step'],d['value'],d['stage_ms']))" "$line")" >> $out
  done
done
cat $out
