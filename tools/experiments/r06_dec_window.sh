#!/bin/bash
# tools/experiments/r06_dec_window.sh -- frame window of the small-batch decode loop (TdtState::F): parity tests on the production library, then
# one 10 s clip end to end and the batch-of-1 stage times with the EXPERIMENTAL build's switch (PK_DEC_WIN=0 / 1), interleaved
export TMPDIR=/tmp
o=gpurun_out/r06_dec_window; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_e2e.py tests/test_gpu_boost.py tests/test_gpu_ragged.py tests/test_gpu_stream.py tests/test_gpu_vs_reference_code.py tests/test_gpu_facade.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
fi
: > $o/ab.txt
for rep in 1 2 3; do for sw in 0 1; do
  echo "dec_win=$sw $(PK_LIB=$exp PK_DEC_WIN=$sw timeout 200 python tools/latency_single.py 2>&1 | tr '\n' ' ')" >> $o/ab.txt
done; done
for rep in 1 2; do for sw in 0 1; do for b in 1 4; do
  line=$(PK_LIB=$exp PK_DEC_WIN=$sw timeout 300 python bench.py --batch $b --no-cpu-baseline --no-also --steps 50 --warmup 5 --sustain-seconds 0 2>/dev/null | tail -1)
  echo "batch=$b dec_win=$sw $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('ms_per_step=%.3f stage_ms=%s parity=%s' % (d['ms_per_step'], d['stage_ms'], d.get('parity', d.get('ids_equal_oracle'))))" "$line")" >> $o/ab.txt
done; done; done
cat $o/ab.txt; tail -5 $o/tests.log
