#!/bin/bash
# tools/experiments/ln_rpw_ab.sh -- rows per wavefront of layernorm_kernel on batches (1 = round 3; 2; 4): LayerNorm ms per step in the bench run, both configurations
out=gpurun_out/ln_rpw_ab.txt
: > $out
for rep in 1 2; do
  for r in 1 2 4; do
    lib=parakeet.cpp_amd/libparakeet_amd_ln$r.so; [ $r = 4 ] && lib=parakeet.cpp_amd/libparakeet_amd.so
    line=$(PK_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "110m rpw=$r $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f layernorm_ms=%.4f (%d launches) enc=%.3f'%(d['ms_per_step'],k['layernorm']['ms'],k['layernorm']['launches'],d['stage_ms']['encoder']))" "$line")" >> $out
  done
done
for r in 1 4; do
  lib=parakeet.cpp_amd/libparakeet_amd_ln$r.so; [ $r = 4 ] && lib=parakeet.cpp_amd/libparakeet_amd.so
  line=$(PK_LIB=$PWD/$lib timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "600m-bf16 rpw=$r $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f layernorm_ms=%.4f (%d launches) enc=%.3f'%(d['ms_per_step'],k['layernorm']['ms'],k['layernorm']['launches'],d['stage_ms']['encoder']))" "$line")" >> $out
done
cat $out
