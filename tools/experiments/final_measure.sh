#!/bin/bash
# round-end measurement set (run on the GPU box through gpurun; every tool call is bounded by its own timeout)
export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
timeout 250 python bench.py > gpurun_out/r02d/bench_v8.json 2> gpurun_out/r02d/bench_v8.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02d/prof -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02d/prof.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02d/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02d/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02d/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02d/pmc_write.log 2>&1
timeout 400 python bench.py --config tdt-600m --bf16 --steps 8 --warmup 2 > gpurun_out/r02d/bench_600m_bf16.json 2> gpurun_out/r02d/bench_600m_bf16.err
ls -la gpurun_out/r02d gpurun_out/r02d/prof | head -40
