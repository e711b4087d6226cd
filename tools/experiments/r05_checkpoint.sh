#!/bin/bash
# tools/experiments/r05_checkpoint.sh TAG -- the driver's round-end sequence on the production library: GPU test suite, smoke(), default bench line
export TMPDIR=/tmp
o=gpurun_out/${1:-r05_checkpoint}
mkdir -p $o
timeout 1200 python -m pytest tests -m gpu -q -x > $o/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $o/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?" >> $o/smoke.log
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?" >> $o/bench.err
tail -3 $o/gpu_tests.log; tail -2 $o/smoke.log; tail -1 $o/bench.err; head -c 300 $o/bench.json
