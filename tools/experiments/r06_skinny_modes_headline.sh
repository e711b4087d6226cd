export TMPDIR=/tmp
o=gpurun_out/r06_modes; mkdir -p $o
A=$PWD/parakeet.cpp_amd/libparakeet_amd_prev2.so; B=$PWD/parakeet.cpp_amd/libparakeet_amd.so
: > $o/hl.txt
for rep in 1 2 3 4 5 6; do for l in A B; do
  lib=$A; [ $l = B ] && lib=$B
  echo "headline lib=$l $(PK_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --steps 30 --warmup 5 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")" >> $o/hl.txt
done; done
cat $o/hl.txt
