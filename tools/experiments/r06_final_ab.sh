#!/bin/bash
# tools/experiments/r06_final_ab.sh -- library of commit 6a10b25 (A: parakeet.cpp_amd/libparakeet_amd_prev.so) against the current one (B), interleaved: headline step,
# configs[2], streaming chunk in both modes
export TMPDIR=/tmp
o=gpurun_out/r06_final_ab; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so
: > $o/ab.txt
for rep in 1 2 3; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  line=$(PK_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "headline lib=$l $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f enc=%.3f attention=%.3f layernorm=%.3f decide=%.3f' % (d['ms_per_step'], d['stage_ms']['encoder'], k['relpos_attention']['ms'], k['layernorm']['ms'], k['tdt_decide']['ms']))" "$line")" >> $o/ab.txt
done; done
for rep in 1 2; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  line=$(PK_LIB=$lib timeout 300 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "600m-bf16 lib=$l $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('ms_per_step=%.3f stage_ms=%s' % (d['ms_per_step'], d['stage_ms']))" "$line")" >> $o/ab.txt
done; done
for rep in 1 2 3; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "stream-bf16 lib=$l $(PK_LIB=$lib timeout 200 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['latency_ms_median'], d['latency_ms_p95'])")" >> $o/ab.txt
  echo "stream-fp32 lib=$l $(PK_LIB=$lib timeout 200 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['latency_ms_median'], d['latency_ms_p95'])")" >> $o/ab.txt
done; done
cat $o/ab.txt
