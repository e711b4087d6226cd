export TMPDIR=/tmp
o=gpurun_out/r06_dec_window; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/chunk.txt
for rep in 1 2; do for ch in 4 8 12 16 32; do
  echo "dec_win=1 chunk=$ch $(PK_LIB=$exp PK_DEC_CHUNK=$ch timeout 200 python tools/latency_single.py 2>&1 | head -1)" >> $o/chunk.txt
done; done
cat $o/chunk.txt
