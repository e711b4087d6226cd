#!/bin/bash
# round 5: the block chain of a steady-state streaming chunk as a hipGraph (EXPERIMENTAL build: PK_STREAM_GRAPH=0 / 1; off in production): interleaved A/B in both modes, parity tests.
o=gpurun_out/r05_stream_graph; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/ab.txt
for rep in 1 2 3; do
  for gsw in 0 1; do
    for mode in "--bf16" ""; do
      line=$(PK_LIB=$exp PK_STREAM_GRAPH=$gsw timeout 200 python tools/bench_stream.py $mode --chunks 120 --warmup 45 2>/dev/null | tail -1)
      echo "graph=$gsw mode=${mode:-fp32} $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('median_ms=%.3f p95=%.3f rtfx=%.0f' % (d['latency_ms_median'], d['latency_ms_p95'], d['aggregate_rtfx']))" "$line" 2>/dev/null)" >> $o/ab.txt
    done
  done
done
cat $o/ab.txt
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_sortformer.py -m gpu -x -q 2>&1 | tail -3 > $o/tests.txt; cat $o/tests.txt
