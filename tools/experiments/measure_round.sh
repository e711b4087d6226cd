#!/bin/bash
# tools/experiments/measure_round.sh TAG -- the evidence set of a round, one gpurun call: default bench line, rocprofv3 --kernel-trace --stats
# of the SAME command, the two --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs: MI355X_MICROARCH.md section HBM), the same for configs[2]
# (tdt-600m, bf16).  Summaries (what gets committed under profiles/) are written next to the raw output in gpurun_out/TAG/.
export TMPDIR=/tmp
tag=${1:-rXX}
o=gpurun_out/$tag
mkdir -p $o
timeout 400 python bench.py > $o/bench.json 2> $o/bench.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0 > $o/prof.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof/*/kt_kernel_trace.csv $o/prof/kt_kernel_trace.csv 2>/dev/null | head -1) $o/kernel_stats.md > /dev/null 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0 > $o/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0 > $o/pmc_write.log 2>&1
python tools/pmc_hbm.py $o/pmc_fetch $o/pmc_write $o/pmc_hbm.json > /dev/null 2>&1
# configs[2]
timeout 400 python bench.py --config tdt-600m --bf16 > $o/bench_600m_bf16.json 2> $o/bench_600m_bf16.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof600 -o kt -- python bench.py --config tdt-600m --bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0 > $o/prof600.log 2>&1
python tools/rocprof_summary.py $(ls $o/prof600/*/kt_kernel_trace.csv $o/prof600/kt_kernel_trace.csv 2>/dev/null | head -1) $o/kernel_stats_600m_bf16.md > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc600_fetch -o f -- python bench.py --config tdt-600m --bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0 > $o/pmc600_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc600_write -o w -- python bench.py --config tdt-600m --bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-also --sustain-seconds 0 > $o/pmc600_write.log 2>&1
python tools/pmc_hbm.py $o/pmc600_fetch $o/pmc600_write $o/pmc_hbm_600m_bf16.json "gemm_bf16_glds_kernel<4, 2, 2, 4, 2" 12032 4096 2 > /dev/null 2>&1
# raw traces are large: keep the summaries only
rm -rf $o/prof $o/prof600 $o/pmc_fetch $o/pmc_write $o/pmc600_fetch $o/pmc600_write
ls -la $o
head -c 600 $o/bench.json; echo; head -30 $o/kernel_stats.md
