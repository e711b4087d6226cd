export TMPDIR=/tmp
o=gpurun_out/r06_modes; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev2.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_e2e.py tests/test_gpu_ragged.py tests/test_gpu_stream.py tests/test_gpu_boost.py tests/test_gpu_vs_reference_code.py tests/test_gpu_group.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do for n in 1 8; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "clips=$n lib=$l $(PK_LIB=$lib PK_LAT_CLIPS=$n timeout 200 python tools/latency_single.py 2>&1 | head -1)" >> $o/ab.txt
done; done; done
for rep in 1; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "headline lib=$l $(PK_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['ms_per_step'], k['lstm_hh_cell']['ms'], k['joint_pred_act']['ms'], k['joint_heads_gemv']['ms'])")" >> $o/ab.txt
done; done
cat $o/ab.txt; tail -3 $o/tests.log
