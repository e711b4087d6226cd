#!/bin/bash
# round-3 library (commit 2a3195b) vs the current one, interleaved on one box: headline and configs[2]
out=gpurun_out/r03_vs_r04_ab.txt
: > $out
for rep in 1 2 3; do
  for lib in parakeet.cpp_amd/libparakeet_amd_prev.so parakeet.cpp_amd/libparakeet_amd.so; do
    line=$(PK_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "110m $lib $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f mel=%.3f enc=%.3f dec=%.3f | '%(d['ms_per_step'],d['stage_ms']['mel'],d['stage_ms']['encoder'],d['stage_ms']['decode'])+' '.join('%s=%.3f'%(n,k[n]['ms']) for n in ('tdt_decide','joint_heads_gemv','lstm_hh_cell','joint_pred_act','relpos_attention','layernorm','dwconv_bn_silu','sub_conv1_dw1','sub_dw2','mel_logmel')))" "$line")" >> $out
  done
done
for rep in 1 2; do
  for lib in parakeet.cpp_amd/libparakeet_amd_prev.so parakeet.cpp_amd/libparakeet_amd.so; do
    line=$(PK_LIB=$PWD/$lib timeout 200 python bench.py --config tdt-600m --bf16 --no-cpu-baseline --no-also --steps 10 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1)
    echo "600m-bf16 $lib $(python -c "import json,sys; d=json.loads(sys.argv[1]); k=d['kernels']; print('ms_per_step=%.3f enc=%.3f dec=%.3f | '%(d['ms_per_step'],d['stage_ms']['encoder'],d['stage_ms']['decode'])+' '.join('%s=%.3f'%(n,k[n]['ms']) for n in ('tdt_decide','joint_heads_gemv','lstm_hh_cell','joint_pred_act','relpos_attention','layernorm')))" "$line")" >> $out
  done
done
cat $out
