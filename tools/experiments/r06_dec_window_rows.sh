#!/bin/bash
# tools/experiments/r06_dec_window_rows.sh -- which lock-step batch sizes gain from the decode frame window: pk_transcribe_pcm of 1 / 2 / 4 / 8 clips per call,
# EXPERIMENTAL build, PK_DEC_WIN = largest batch that gets a window (0: off), interleaved, median of 100 calls
export TMPDIR=/tmp
o=gpurun_out/r06_dec_window; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/rows.txt
for rep in 1 2; do for n in 1 2 4 8; do for sw in 0 2 4 8; do
  echo "clips=$n dec_win=$sw $(PK_LIB=$exp PK_LAT_CLIPS=$n PK_DEC_WIN=$sw timeout 200 python tools/latency_single.py 2>&1 | head -1)" >> $o/rows.txt
done; done; done
cat $o/rows.txt
