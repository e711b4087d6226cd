#!/bin/bash
# tools/experiments/r05_sq.sh -- the SQ counter evidence of round 5 (one rocprofv3 --pmc pass each: 8 SQ counters + GRBM_GUI_ACTIVE, no
# trace domains beside it): configs[1] headline, configs[2] bf16 (LDS epilogue / persistent + direct epilogue / continuous stream),
# configs[4] streaming in both modes.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).
export TMPDIR=/tmp
o=gpurun_out/r05_sq
mkdir -p $o
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE"
run() { # tag title cmd...
  tag=$1; title=$2; shift 2
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $o/$tag -o sq -- "$@" > $o/$tag.log 2>&1
  python tools/pmc_sq_summary.py $o/$tag $o/sq_$tag.md "# rocprofv3 --pmc (8 SQ counters + GRBM_GUI_ACTIVE), $title.  Mean per dispatch." > /dev/null 2>&1
  rm -rf $o/$tag
}
B="--steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0"
run 110m "\`python bench.py $B\` (configs[1], fp32)" python bench.py $B
for p in 0 2 4; do
  PK_LIB=$exp PK_BF16_PERSIST=$p run 600m_bf16_p$p "\`PK_BF16_PERSIST=$p python bench.py --config tdt-600m --bf16 $B\` (configs[2]; EXPERIMENTAL build)" python bench.py --config tdt-600m --bf16 $B
done
run stream_fp32 "\`python tools/bench_stream.py --chunks 30 --warmup 5\` (configs[4], exact mode)" python tools/bench_stream.py --chunks 30 --warmup 5
run stream_bf16 "\`python tools/bench_stream.py --bf16 --chunks 30 --warmup 5\` (configs[4], tolerance-class mode)" python tools/bench_stream.py --bf16 --chunks 30 --warmup 5
ls -la $o; head -12 $o/sq_600m_bf16_p2.md
