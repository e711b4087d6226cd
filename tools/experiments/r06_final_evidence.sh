#!/bin/bash
# tools/experiments/r06_final_evidence.sh TAG -- the evidence set of round 6 on the PRODUCTION library (same set as round 5's + the mixed-length distributions the
# round-5 verdict asked to re-measure): measure_round.sh (bench line, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes for configs[1] and configs[2]) + one SQ
# counter pass each for configs[1], configs[2] bf16 and configs[4] in both modes + the streaming kernel tables + teacher-forced parity figures.
#   gpurun -- bash tools/experiments/r06_final_evidence.sh r06_m2
export TMPDIR=/tmp
tag=${1:-r06_m2}
bash tools/experiments/measure_round.sh $tag > gpurun_out/${tag}_measure.log 2>&1
o=gpurun_out/$tag
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE"
run() { # tag title cmd...
  t=$1; title=$2; shift 2
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $o/$t -o sq -- "$@" > $o/$t.log 2>&1
  python tools/pmc_sq_summary.py $o/$t $o/pmc_sq_$t.md "# rocprofv3 --pmc (8 SQ counters + GRBM_GUI_ACTIVE), $title.  Mean per dispatch." > /dev/null 2>&1
  rm -rf $o/$t $o/$t.log
}
B="--steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0"
run 110m "\`python bench.py $B\` (configs[1], fp32)" python bench.py $B
run 600m_bf16 "\`python bench.py --config tdt-600m --bf16 $B\` (configs[2], production library)" python bench.py --config tdt-600m --bf16 $B
run stream_fp32 "\`python tools/bench_stream.py --chunks 30 --warmup 5\` (configs[4], exact mode)" python tools/bench_stream.py --chunks 30 --warmup 5
run stream_bf16 "\`python tools/bench_stream.py --bf16 --chunks 30 --warmup 5\` (configs[4], tolerance-class mode)" python tools/bench_stream.py --bf16 --chunks 30 --warmup 5
for m in fp32 bf16; do
  f=""; [ $m = bf16 ] && f="--bf16"
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/ps_$m -o kt -- python tools/bench_stream.py $f --chunks 45 --warmup 5 > $o/ps_$m.log 2>&1
  python tools/rocprof_summary.py $(ls $o/ps_$m/*/kt_kernel_trace.csv $o/ps_$m/kt_kernel_trace.csv 2>/dev/null | head -1) $o/stream_${m}_kernel_stats.md > /dev/null 2>&1
  python tools/stream_kernel_tbps.py $o/stream_${m}_kernel_stats.md $m > $o/stream_${m}_kernel_tbps.md 2>/dev/null
  rm -rf $o/ps_$m $o/ps_$m.log
done
timeout 300 python tools/bench_stream.py --chunks 100 --warmup 10 2>/dev/null | tail -1 > $o/bench_stream_fp32.json
timeout 300 python tools/bench_stream.py --bf16 --chunks 100 --warmup 10 2>/dev/null | tail -1 > $o/bench_stream_bf16.json
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_600m_depth.py -m gpu -q -s -k "teacher_forced or fp32_reference_path or stream_score" 2>&1 | grep -v "^$" | tail -40 > $o/teacher_forced_tests.txt
# mixed-length distributions on the final code (round-5 verdict item 3: re-measure profiles/r04_mixed_bench_distributions.txt)
: > $o/mixed_distributions.txt
for d in "1 5" "5 15" "2 30" "20 30"; do
  set -- $d
  echo "== clips uniform in [$1, $2] s: $(timeout 400 python tools/bench_mixed.py --lo $1 --hi $2 --clips 256 --oracle-sample 2 2>/dev/null | tail -1)" >> $o/mixed_distributions.txt
done
timeout 300 python tools/latency_single.py > $o/latency_single.txt 2>&1
timeout 600 python tools/bench_reference_protocol.py > $o/reference_protocol.json 2> /dev/null
ls -la $o; head -c 400 $o/bench.json; echo; head -8 $o/pmc_sq_600m_bf16.md; cat $o/pmc_hbm.json | head -c 600; echo; cat $o/pmc_hbm_600m_bf16.json | head -c 600
