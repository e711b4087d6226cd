#!/usr/bin/env python3
"""tools/bench_mixed.py -- RTFx of the hot path on MIXED-LENGTH input next to the equal-length headline (bench.py), one GPU.

The reference's roadmap item "batch inference: pad + length-mask" (README.md:513) is implemented by PACKING (include/parakeet_amd.h, "ragged"
entry points): clips of different lengths share one batch, no padded frame is computed.  This tool measures what that is worth:

  equal      64 x 10 s, uniform pipeline (the headline configuration of bench.py)
  mixed      the same amount of audio as clips of seeded random lengths (default uniform 5-15 s; --lo / --hi), ONE ragged batch per step,
             resident in HBM like bench.py's batch; decode groups as in bench.py
  one_call   pk_transcribe_pcm on --clips clips of such lengths from host memory (sorting, packing, PCIe uploads, results: all inside)
  serial     what the round-3 engine did with mixed lengths: every length class alone (here: every clip alone), for scale

and checks, on the timed mixed batch, that every clip's tokens equal its single-clip transcription (and the CPU oracle's on a sample).
Prints one JSON line; `python tools/bench_mixed.py > profiles/rNN_mixed_bench.json`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

import bench  # noqa: E402  (weights_file, log)


def timed_steps(L, capi, batch, dec, steps, warmup):
    for _ in range(warmup):
        capi.check(L.pk_batch_run(batch, dec))
    capi.check(L.pk_batch_sync(batch))
    t0 = time.perf_counter()
    for _ in range(steps):
        capi.check(L.pk_batch_run(batch, dec))
    capi.check(L.pk_batch_sync(batch))
    return (time.perf_counter() - t0) / steps


def kernel_ms(L, capi, batch, dec, n=3):
    """per-kernel HIP-event milliseconds of one un-pipelined step (pk_batch_profile), averaged over n steps; decode-loop kernels summed"""
    agg = {}
    for _ in range(n):
        stats = (capi.PkKernelStat * 64)()
        nk = L.pk_batch_profile(batch, dec, stats, 64)
        for i in range(max(0, min(nk, 64))):
            agg[stats[i].name.decode()] = agg.get(stats[i].name.decode(), 0.0) + stats[i].total_ms / n
    return {k: round(v, 3) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--lo", type=float, default=5.0, help="shortest clip, seconds")
    ap.add_argument("--hi", type=float, default=15.0, help="longest clip, seconds")
    ap.add_argument("--audio", type=float, default=660.0, help="upper bound on the seconds of audio per mixed batch (the row budget usually binds first)")
    ap.add_argument("--clips", type=int, default=512, help="clips of the one-call measurement")
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--oracle-sample", type=int, default=4)
    ap.add_argument("--seed", type=int, default=77)
    args = ap.parse_args()

    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import capi, synth
    if capi.device_count() < 1:
        raise SystemExit("bench_mixed.py needs an MI355X: the product has no CPU path")
    cfg = pk.make_110m_config()
    wpath, W = bench.weights_file(cfg)
    model = capi.Model(wpath, cfg, device=0)
    L = capi.lib()
    dec = 1
    rng = np.random.default_rng(args.seed)

    # ---- equal: bench.py's configuration --------------------------------------------------------------------------------------------
    eq = C.c_void_p()
    capi.check(L.pk_batch_create(model._h, 64, 160000, C.byref(eq)))
    pcm_eq = synth.synth_pcm(64, 160000, seed=1234)
    capi.check(L.pk_batch_upload(eq, pcm_eq.ctypes.data_as(capi.f32p), 64))
    capi.check(L.pk_batch_set_decode_group(eq, args.group))
    s_eq = timed_steps(L, capi, eq, dec, args.steps, args.warmup)
    L.pk_batch_free(eq)
    rtfx_eq = 640.0 / s_eq

    # ---- mixed: one ragged batch holding the same amount of audio ------------------------------------------------------------------------
    # the batch the one-call API would form from a stream of such clips: clips are added until the next one would exceed the packing budget of
    # 8192 encoder rows (pk_plan_batches; --audio caps the seconds as well)
    lens, rows = [], 0
    while True:
        n = int(rng.uniform(args.lo, args.hi) * 16000)
        r = L.pk_encoder_num_frames(L.pk_mel_num_frames(n))
        if rows + r > 8192 or (sum(lens) + n) / 16000.0 > args.audio + args.hi or len(lens) >= 256:
            break
        lens.append(n); rows += r
    clips = [synth.synth_pcm(1, n, seed=9000 + i)[0] for i, n in enumerate(lens)]
    audio_s = sum(lens) / 16000.0
    bt = capi.Batch.ragged(model, len(clips), sum(lens), max(lens))
    bt.upload_ragged(clips)
    bt.set_decode_group(args.group)
    s_mx = timed_steps(L, capi, bt._h, dec, args.steps, args.warmup)
    res = bt.results()
    rtfx_mx = audio_s / s_mx
    ms = np.zeros(4, np.float32)
    capi.check(L.pk_batch_run_timed(bt._h, dec, ms.ctypes.data_as(capi.f32p)))
    kern_mx = kernel_ms(L, capi, bt._h, dec)
    bt.close()
    eq2 = C.c_void_p()
    capi.check(L.pk_batch_create(model._h, 64, 160000, C.byref(eq2)))
    capi.check(L.pk_batch_upload(eq2, pcm_eq.ctypes.data_as(capi.f32p), 64))
    kern_eq = kernel_ms(L, capi, eq2, dec)
    L.pk_batch_free(eq2)

    # parity of the timed batch: every clip vs its single-clip transcription; a sample vs the CPU oracle
    mism = 0
    t0 = time.perf_counter()
    alone = [model.transcribe_pcm([c], decoder="tdt")[0]["token_ids"] for c in clips]
    s_serial = time.perf_counter() - t0
    for i in range(len(clips)):
        if res["ids"][i, :res["lens"][i]].tolist() != alone[i]:
            mism += 1
    o_mism, o_n = 0, 0
    if args.oracle_sample > 0:
        import oracle
        oracle.set_threads(min(8, os.cpu_count() or 1))
        om = oracle.Model(cfg, W if W is not None else synth.synth_weights(cfg, seed=42))
        order = np.argsort(lens)
        pick = sorted(set([int(order[0]), int(order[-1])] + [int(order[(k + 1) * len(lens) // (args.oracle_sample + 1)]) for k in range(args.oracle_sample - 2)]))
        for i in pick:
            o = om.tdt_greedy(om.encoder(oracle.mel(clips[i])[None]))
            o_n += 1
            if o["ids"][0, :o["lens"][0]].tolist() != res["ids"][i, :res["lens"][i]].tolist():
                o_mism += 1

    # ---- one call from host memory --------------------------------------------------------------------------------------------------------
    lens2 = [int(rng.uniform(args.lo, args.hi) * 16000) for _ in range(args.clips)]
    clips2 = [synth.synth_pcm(1, n, seed=20000 + i)[0] for i, n in enumerate(lens2)]
    model.transcribe_pcm(clips2[:8], decoder="tdt")                      # warm the pipeline's allocations
    packed2 = capi.pack_clips(clips2)                                    # pk_transcribe_pcm's input form (pcm + offsets)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r2 = model.transcribe_pcm(packed2, decoder="tdt")
        best = min(best, time.perf_counter() - t0)
    audio2 = sum(lens2) / 16000.0
    model.close()

    line = {
        "workload": f"tdt-ctc-110m, fp32, TDT greedy; mixed-length clips uniform {args.lo:g}-{args.hi:g} s (seed {args.seed}) packed into ragged batches vs 64 x 10 s",
        "equal_length": {"ms_per_step": round(s_eq * 1e3, 3), "rtfx": round(rtfx_eq, 1), "clips": 64, "audio_s": 640.0},
        "mixed_resident": {"ms_per_step": round(s_mx * 1e3, 3), "rtfx": round(rtfx_mx, 1), "clips": len(clips), "audio_s": round(audio_s, 2), "encoder_rows": int(rows),
                           "shortest_s": round(min(lens) / 16000, 2), "longest_s": round(max(lens) / 16000, 2), "decode_group": args.group,
                           "stage_ms_unpipelined": {"mel": round(float(ms[0]), 3), "encoder": round(float(ms[1]), 3), "decode": round(float(ms[2]), 3)}},
        "mixed_over_equal": round(rtfx_mx / rtfx_eq, 4),
        "kernel_ms_unpipelined": {k: {"equal": kern_eq.get(k), "mixed": kern_mx.get(k)} for k in sorted(set(kern_eq) | set(kern_mx))},
        "one_call_from_host": {"clips": args.clips, "audio_s": round(audio2, 1), "wall_s": round(best, 4), "rtfx": round(audio2 / best, 1),
                               "note": "pk_transcribe_pcm: sort, pack (<= 256 clips / 8192 encoder rows per batch), PCIe uploads, pipeline, results, detokenise-free; best of 3"},
        "one_clip_at_a_time": {"clips": len(clips), "wall_s": round(s_serial, 4), "rtfx": round(audio_s / s_serial, 1),
                               "note": "the same mixed clips, one pk_transcribe_pcm call per clip: what every length class cost before packing"},
        "parity": {"clips": len(clips), "token_mismatches_vs_single_clip": mism, "oracle_clips": o_n, "token_mismatches_vs_oracle": o_mism},
    }
    print(json.dumps(line), flush=True)
    if mism or o_mism:
        raise SystemExit(2)


if __name__ == "__main__":
    main()
