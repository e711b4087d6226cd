export TMPDIR=/tmp
o=gpurun_out/r06_dec2; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_e2e.py tests/test_gpu_boost.py tests/test_gpu_ragged.py tests/test_gpu_stream.py tests/test_gpu_vs_reference_code.py tests/test_gpu_facade.py tests/test_gpu_bf16.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
bash tools/experiments/r06_single_profile.sh > $o/profile.txt 2>&1
: > $o/ab.txt
for rep in 1 2; do for sw in 0 2 4 8; do for b in 1 2 4 8; do
  line=$(PK_LIB=$exp PK_DEC_WIN=$sw timeout 300 python bench.py --batch $b --no-cpu-baseline --no-also --steps 50 --warmup 5 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "batch=$b dec_win=$sw $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('ms_per_step=%.3f stage_ms=%s' % (d['ms_per_step'], d['stage_ms']))" "$line")" >> $o/ab.txt
done; done; done
grep "decide\|skinny\|clip end" $o/profile.txt; cat $o/ab.txt; tail -3 $o/tests.log
