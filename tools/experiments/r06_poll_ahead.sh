#!/bin/bash
# tools/experiments/r06_poll_ahead.sh -- the decode loop's poll two steps ahead of the chunk's end, waited for by event (run_tdt_loop): tests on the production
# library, then EXPERIMENTAL build with PK_DEC_POLL_AHEAD=0 / 1, interleaved: pk_transcribe_pcm of 1 / 8 clips per call (median of 100 calls), headline step
export TMPDIR=/tmp
o=gpurun_out/r06_poll; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_e2e.py tests/test_gpu_ragged.py tests/test_gpu_stream.py tests/test_gpu_boost.py tests/test_gpu_vs_reference_code.py tests/test_gpu_group.py tests/test_gpu_facade.py -m gpu -q -x > $o/tests.log 2>&1
echo "tests rc=$?" >> $o/tests.log
: > $o/ab.txt
for rep in 1 2 3; do for n in 1 8; do for sw in 0 1; do
  echo "clips=$n poll_ahead=$sw $(PK_LIB=$exp PK_DEC_POLL_AHEAD=$sw PK_LAT_CLIPS=$n timeout 200 python tools/latency_single.py 2>&1 | head -1)" >> $o/ab.txt
done; done; done
for rep in 1 2 3; do for sw in 0 1; do
  echo "headline poll_ahead=$sw $(PK_LIB=$exp PK_DEC_POLL_AHEAD=$sw timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")" >> $o/ab.txt
done; done
cat $o/ab.txt; tail -3 $o/tests.log
