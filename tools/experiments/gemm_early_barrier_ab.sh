#!/bin/bash
# tools/experiments/gemm_early_barrier_ab.sh -- the single-buffered fp32 GEMM loop with its first barrier one sub-step earlier (-DGP_EARLY=1 build:
# four fragment slots, 16 instead of 8 MFMAs between the two barriers of a K tile) against the production loop: parity, then the default bench
# per library, interleaved on one box.  (Round 4: slower, profiles/r04_gemm_early_barrier_ab.txt; the -DGP_EARLY patch was not kept in gemm_pipe.hpp --
# the script documents how the A/B was run.)
#   make -C parakeet.cpp_amd/csrc OBJDIR=build_exp LIB=../libparakeet_amd_exp.so EXTRA_CXXFLAGS=-DGP_EARLY=1
PK_LIB=$PWD/parakeet.cpp_amd/libparakeet_amd_exp.so timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_encoder.py -m gpu -x -q 2>&1 | tail -2
LIBS="parakeet.cpp_amd/libparakeet_amd.so parakeet.cpp_amd/libparakeet_amd_exp.so" BENCH_ARGS=--no-also bash tools/experiments/lib_ab.sh
