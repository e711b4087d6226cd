#!/bin/bash
# tools/experiments/r06_ratio_attribution.sh -- round 6 (round-5 verdict, weak #1 / item 5): the bf16 STREAMING mode's distance from the fp32 reference path, as a
# distribution (mean, percentiles, max) against the bf16 oracle's own, under each summation-order switch of the EXPERIMENTAL build.
export TMPDIR=/tmp
o=gpurun_out/r06_ratio; mkdir -p $o; exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
: > $o/attribution.txt
run() { # label env...
  lab=$1; shift
  echo "== $lab" >> $o/attribution.txt
  env PK_LIB=$exp PK_TEST_FP32_PATH_ONLY=1 "$@" timeout 600 python -m pytest tests/test_gpu_stream.py -m gpu -q -s -k "teacher_forced_bf16" 2>&1 | grep -E "distance from fp32|vs the fp32 oracle|passed|failed|Error" >> $o/attribution.txt
}
run "production switches" PK_DUMMY=0
run "folded final norm OFF (PK_STREAM_FUSE_FIN=0)" PK_STREAM_FUSE_FIN=0
run "folded LayerNorms OFF (PK_STREAM_FUSE_LN=0)" PK_STREAM_FUSE_LN=0
run "LDS-DMA rows OFF (PK_SB_AL=0)" PK_SB_AL=0
run "K slices met in REVERSE wave order (PK_SB_SUMREV=1)" PK_SB_SUMREV=1
run "operand-tile weights OFF (PK_SB_WT=0)" PK_SB_WT=0
run "8-row activation tiles OFF (PK_STREAM_ACT_TILES=0)" PK_STREAM_ACT_TILES=0
run "conv tail OFF (PK_STREAM_FUSE_DW=0)" PK_STREAM_FUSE_DW=0
cat $o/attribution.txt
