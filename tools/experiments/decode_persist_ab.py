#!/usr/bin/env python3
"""Round 4 A/B of the TDT greedy loop's launch structure with the XCD-hierarchical grid barrier in the single-launch kernel
(kernels/decode_persist.hip): per-phase launches (default) vs the persistent loop, tdt-ctc-110m.
  * one 10 s clip end to end (pk_transcribe_pcm) and its decode stage alone;
  * the decode stage of a 64-clip batch alone (pk_batch_run_timed), i.e. per-symbol-step cost at B = 64;
  * the pipelined headline step (bench.py protocol, decode groups of 4, overlapped with the next encoder).
usage: python tools/experiments/decode_persist_ab.py > gpurun_out/decode_persist_ab.txt"""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402
import bench  # noqa: E402
import pkload  # noqa: E402

pk = pkload.load()
from parakeet_cpp_amd import capi, synth  # noqa: E402

cfg = pk.make_110m_config()
wp, _ = bench.weights_file(cfg)
gm = capi.Model(wp, cfg, device=0)
L = capi.lib()
pcm64 = synth.synth_pcm(64, 160000, seed=1234)
one = pcm64[0]
ref_ids = {}
for mode in ("phases", "persistent"):
    gm.set_decode_loop(mode)
    # single clip, end to end
    gm.transcribe_pcm([one], "tdt")
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); r = gm.transcribe_pcm([one], "tdt"); ts.append((time.perf_counter() - t0) * 1e3)
    enc = gm.encode(gm.mel(one[None]))
    gm.tdt_decode(enc)
    td = []
    for _ in range(30):
        t0 = time.perf_counter(); g = gm.tdt_decode(enc); td.append((time.perf_counter() - t0) * 1e3)
    steps = int(g["steps"][0])
    print(f"{mode:10s} one 10 s clip: end to end {statistics.median(ts):.3f} ms; decode stage alone (host buffers) {statistics.median(td):.3f} ms for {steps} symbol steps "
          f"= {statistics.median(td) * 1e3 / steps:.1f} us per step; tokens {len(r[0]['token_ids'])}")
    ref_ids.setdefault("one", r[0]["token_ids"])
    assert r[0]["token_ids"] == ref_ids["one"], "token ids differ between the loop forms"
    # 64 clips: stage timers of an un-pipelined run
    b = C.c_void_p()
    capi.check(L.pk_batch_create(gm._h, 64, 160000, C.byref(b)))
    capi.check(L.pk_batch_upload(b, pcm64.ctypes.data_as(capi.f32p), 64))
    ms = (C.c_float * 4)()
    dec = []
    for _ in range(6):
        capi.check(L.pk_batch_run_timed(b, 1, ms)); dec.append(float(ms[2]))
    mt = L.pk_batch_max_tokens(b)
    ids = np.zeros((64, mt), np.int32); lens = np.zeros(64, np.int32)
    capi.check(L.pk_batch_results(b, ids.ctypes.data_as(capi.i32p), lens.ctypes.data_as(capi.i32p), None, None, None))
    key = [ids[i, :lens[i]].tolist() for i in range(64)]
    ref_ids.setdefault("b64", key)
    assert key == ref_ids["b64"], "token ids differ between the loop forms (batch 64)"
    print(f"{mode:10s} 64 x 10 s, decode stage alone (no encoder beside it): {statistics.median(dec):.3f} ms")
    # pipelined headline step
    for group in (1, 4):
        capi.check(L.pk_batch_set_decode_group(b, group))
        for _ in range(4):
            capi.check(L.pk_batch_run(b, 1))
        capi.check(L.pk_batch_sync(b))
        wins = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                capi.check(L.pk_batch_run(b, 1))
            capi.check(L.pk_batch_sync(b))
            wins.append((time.perf_counter() - t0) / 20 * 1e3)
        print(f"{mode:10s} pipelined step, decode group {group}: {statistics.median(wins):.3f} ms per 64 x 10 s step (min {min(wins):.3f})")
    L.pk_batch_free(b)
gm.close()
