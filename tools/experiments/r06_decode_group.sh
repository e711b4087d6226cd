#!/bin/bash
# tools/experiments/r06_decode_group.sh -- headline step by decode group size on the final code (round 3's table: profiles/r03_decode_modes_ab.txt), CTC-only as the floor
export TMPDIR=/tmp
o=gpurun_out/r06_group; mkdir -p $o; : > $o/ab.txt
for rep in 1 2; do
  for g in 1 2 4 8 16; do
    echo "--decode-group $g | $(timeout 300 python bench.py --decode-group $g --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")" >> $o/ab.txt
  done
  echo "--decoder ctc | $(timeout 300 python bench.py --decoder ctc --no-cpu-baseline --no-also --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")" >> $o/ab.txt
done
cat $o/ab.txt
