export TMPDIR=/tmp
o=gpurun_out/r06_single_loops; mkdir -p $o
for rep in 1 2; do for lp in phases graph persistent; do PK_LAT_DECODE_LOOP=$lp timeout 200 python tools/latency_single.py 2>&1 | tr '\n' ' ' >> $o/loops.txt; echo >> $o/loops.txt; done; done
cat $o/loops.txt
