#!/usr/bin/env python3
"""Experiment (decode overlap): does the mere LAUNCHING of ~500 small kernels per step on a second stream slow the encoder?
Runs encoder-only steps (CTC) of the bench configuration, alone and with a host thread launching tiny torch kernels (a 64-element add)
on another stream at ~25 000 launches / s (512 per 20 ms step, the TDT loop's launch count).  Kernel boundaries carry cache
release / acquire operations; if these cost the encoder, a persistent decode kernel removes them."""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import bench  # noqa: E402
import pkload  # noqa: E402
import torch  # noqa: E402

pk = pkload.load()
from parakeet_cpp_amd import capi, synth  # noqa: E402

cfg = pk.make_110m_config()
wpath, _ = bench.weights_file(cfg)
model = capi.Model(wpath, cfg, device=0)
L = capi.lib()
batch = C.c_void_p()
capi.check(L.pk_batch_create(model._h, 64, 160000, C.byref(batch)))
pcm = synth.synth_pcm(64, 160000, seed=1234)
capi.check(L.pk_batch_upload(batch, pcm.ctypes.data_as(capi.f32p), 64))


def steps(n, dec):
    for _ in range(3):
        capi.check(L.pk_batch_run(batch, dec))
    capi.check(L.pk_batch_sync(batch))
    t0 = time.perf_counter()
    for _ in range(n):
        capi.check(L.pk_batch_run(batch, dec))
    capi.check(L.pk_batch_sync(batch))
    return (time.perf_counter() - t0) / n * 1e3


stop = False
count = [0]


def spam(rate):
    s = torch.cuda.Stream()
    x = torch.zeros(64, device="cuda")
    period = 1.0 / rate
    nxt = time.perf_counter()
    with torch.cuda.stream(s):
        while not stop:
            x.add_(1.0)
            count[0] += 1
            nxt += period
            d = nxt - time.perf_counter()
            if d > 0:
                time.sleep(d)


print("encoder only (CTC), alone        : %.3f ms / step" % steps(40, 0))
print("encoder + TDT decode overlapped  : %.3f ms / step" % steps(40, 1))
for rate in (25000, 100000):
    stop = False
    count[0] = 0
    th = threading.Thread(target=spam, args=(rate,))
    th.start()
    time.sleep(0.2)
    c0, t0 = count[0], time.perf_counter()
    ms = steps(40, 0)
    c1, t1 = count[0], time.perf_counter()
    stop = True
    th.join()
    print("encoder only + %.0f tiny launches / s on another stream (%.0f per step): %.3f ms / step" % ((c1 - c0) / (t1 - t0), (c1 - c0) / (t1 - t0) * ms / 1e3, ms))
