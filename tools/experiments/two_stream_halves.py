#!/usr/bin/env python3
"""Experiment: does running two half-batches on two streams (two model instances) beat one full batch?  Encoder + CTC only."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import pkload
pk = pkload.load()
from parakeet_cpp_amd import capi, synth
import bench
cfg = pk.make_110m_config()
path, _ = bench.weights_file(cfg)
L = capi.lib()
def mk(B, pcm):
    m = capi.Model(path, cfg, device=0)
    b = C.c_void_p()
    capi.check(L.pk_batch_create(m._h, B, 160000, C.byref(b)))
    capi.check(L.pk_batch_upload(b, np.ascontiguousarray(pcm).ctypes.data_as(capi.f32p), B))
    return m, b
pcm = synth.synth_pcm(64, 160000, seed=1234)
m0, b0 = mk(64, pcm)
m1, b1 = mk(32, pcm[:32]); m2, b2 = mk(32, pcm[32:])
def run_full(n):
    for _ in range(n): capi.check(L.pk_batch_run(b0, 0))
    capi.check(L.pk_batch_sync(b0))
def run_halves(n):
    for _ in range(n):
        capi.check(L.pk_batch_run(b1, 0)); capi.check(L.pk_batch_run(b2, 0))
    capi.check(L.pk_batch_sync(b1)); capi.check(L.pk_batch_sync(b2))
for name, f in (("full 64", run_full), ("2 x 32 on two streams", run_halves), ("full 64", run_full), ("2 x 32 on two streams", run_halves)):
    f(3)
    t = time.perf_counter(); f(10); dt = (time.perf_counter() - t) / 10
    print(f"{name}: {dt*1e3:.2f} ms per 64 clips (mel+encoder+CTC)")
