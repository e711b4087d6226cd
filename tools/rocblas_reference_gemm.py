#!/usr/bin/env python3
"""Context for the GEMM numbers in DESIGN.md: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS, fp32, TF32 off) reaches
on the encoder's product shapes on the same MI355X.  Plain products only (no fused bias / SiLU / residual epilogue, any summation
order) -- an upper-bound style comparison for the bit-exact, epilogue-fused kernels of gemm_pipe.hpp."""
import json
import time

import torch

torch.backends.cuda.matmul.allow_tf32 = False
shapes = {"fc1 8064x2048x512": (8064, 2048, 512), "fc2 8064x512x2048": (8064, 512, 2048), "qkv 8064x1536x512": (8064, 1536, 512),
          "out 8064x512x512": (8064, 512, 512), "sub_pw 321280x256x256": (321280, 256, 256), "600m fc1 12032x4096x1024": (12032, 4096, 1024)}
out = {}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    for _ in range(5):
        (a @ w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        (a @ w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    out[name] = {"us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}
# the same for the bf16 products of configs[2] (bf16 operands, fp32 accumulation inside the library, bf16 output)
for name, (M, N, K) in {"bf16 600m fc1 12032x4096x1024": (12032, 4096, 1024), "bf16 600m qkv 12032x3072x1024": (12032, 3072, 1024),
                        "bf16 600m fc2 12032x1024x4096": (12032, 1024, 4096), "bf16 600m out 12032x1024x1024": (12032, 1024, 1024)}.items():
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(5):
        (a @ w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        (a @ w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    out[name] = {"us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}
print(json.dumps(out))
