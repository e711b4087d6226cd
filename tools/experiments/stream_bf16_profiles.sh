#!/bin/bash
# tools/experiments/stream_bf16_profiles.sh TAG -- rocprofv3 --kernel-trace --stats of the streaming bench in its three forms (fp32 exact, bf16 with the
# folded LayerNorm, bf16 with plain LayerNorm launches: EXPERIMENTAL build, PK_STREAM_FUSE_LN=0) on one box.  Output: gpurun_out/TAG/
o=gpurun_out/${1:-stream_prof}
mkdir -p $o
export TMPDIR=/tmp
exp=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so
prof() {   # name, env..., -- args
  name=$1; shift
  env "$@" timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$name -o kt -- python tools/bench_stream.py --chunks 45 --warmup 5 $EXTRA > $o/prof_$name.log 2>&1
  python tools/rocprof_summary.py $(ls $o/prof_$name/*/kt_kernel_trace.csv $o/prof_$name/kt_kernel_trace.csv 2>/dev/null | head -1) $o/${name}_kernel_stats.md > /dev/null 2>&1
  rm -rf $o/prof_$name
}
EXTRA="" prof fp32 PK_NONE=1
EXTRA="--bf16" prof bf16_fused PK_LIB=$exp PK_STREAM_FUSE_LN=1
EXTRA="--bf16" prof bf16_plain PK_LIB=$exp PK_STREAM_FUSE_LN=0
for n in fp32 bf16_fused bf16_plain; do echo "== $n"; head -14 $o/${n}_kernel_stats.md; done
