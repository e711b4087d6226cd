#!/usr/bin/env python3
"""Generate tests/golden/nemotron600m_stream_score_depth24_seed42.npz: the TEACHER-FORCED joint scores of BASELINE configs[4] (nemotron-600m
streaming, 24 layers, att_context 70 / 1, 160 ms chunks) chunk by chunk along the CPU oracle's own decision paths, state carried across
chunks -- the logits-level parity statement of the streaming modes (round-4 verdict, item 1b: the token contract of the bf16 streaming mode
covered 53 tokens on two streams and ended at the first near-tie).

Sessions: N_STREAMS x N_CHUNKS chunks of 2560 samples of synth_pcm(N_STREAMS, ..., seed 4242); synth_weights(nemotron-600m, seed 42).
Per stream three oracle walks (oracle.Stream: StreamingAudioPreprocessor + forward_chunk + rnnt_streaming_decode_chunk; reference
src/audio.cpp:171-259, src/streaming_encoder.cpp:430-472, src/eou.cpp:17-98):

  A  fp32 oracle, greedy                         -> f32_*   (the reference's arithmetic: the path and scores everything is measured against)
  B  bf16-mode oracle (gemm_bf16), greedy        -> b16_*   (the specification of the tolerance-class mode)
  C  bf16-mode oracle teacher-forced along A     -> b16_on_f32_*  (the mode's own distance from fp32 at the logits)

Per chunk i, stream s, step k (arrays [N_CHUNKS][N_STREAMS][STEP_CAP], n = *_n[i][s] valid steps):
  labels / dur_idx      the decision of every step of the chunk's loop (blank steps included; duration as an index into cfg.durations)
  top_ids / top_lp      the K = 8 most probable labels of the step and their log-probs                          [..][8]
  dur_lp                the duration head's log-probs                                                           [..][D]
  margin                top-1 minus top-2 label log-prob, and the duration head's                               [..][2]
  f32_row_xor / f32_row_sum   fp32 only: xor and uint64 sum of the bit patterns of the whole label log-prob row
  f32_enc_xor / f32_enc_sum   fp32 only, per chunk and stream: checksums of the chunk's encoder output bits
  enc_n                 encoder frames of the chunk (0: buffered)
tests/test_gpu_stream.py walks these paths on the GPU with pk_stream_score: fp32 rows bit-identical along A; bf16 mode within a stated bound
of B along B, and its distance from A (along A) held against C's.
usage (authoring container, ~25 min of CPU on 8 threads): python tools/make_golden_stream_600m_score.py"""
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
OUT = os.path.join(ROOT, "tests", "golden", "nemotron600m_stream_score_depth24_seed42.npz")
N_STREAMS, N_CHUNKS, CHUNK, PCM_SEED, LEFT, RIGHT, K = 8, 80, 2560, 4242, 70, 1, 8


def main():
    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import synth
    import oracle
    threads = min(8, os.cpu_count() or 1)
    oracle.set_threads(threads)
    cfg = pk.make_nemotron_600m_config()
    n_chunks = int(os.environ.get("PK_SCORE_CHUNKS", N_CHUNKS))
    W = synth.synth_weights(cfg, seed=42)
    pcm = synth.synth_pcm(N_STREAMS, CHUNK * N_CHUNKS, seed=PCM_SEED)
    S, D, V = N_STREAMS, len(cfg.durations), cfg.vocab_size
    STEP_CAP = 2 * (cfg.max_symbols_per_step + 1) + 16
    out = {"n_streams": S, "n_chunks": n_chunks, "chunk": CHUNK, "pcm_seed": PCM_SEED, "weights_seed": 42, "att_left": LEFT, "att_right": RIGHT,
           "top_k": K, "step_cap": STEP_CAP, "pcm_chunks": N_CHUNKS, "pcm_digest": np.asarray(pcm, np.float64).sum(axis=1), "durations": np.asarray(cfg.durations, np.int32)}

    def arrays(prefix, rows):
        a = {"n": np.zeros((n_chunks, S), np.int32), "labels": np.full((n_chunks, S, STEP_CAP), -1, np.int32),
             "dur_idx": np.full((n_chunks, S, STEP_CAP), -1, np.int32), "top_ids": np.full((n_chunks, S, STEP_CAP, K), -1, np.int32),
             "top_lp": np.zeros((n_chunks, S, STEP_CAP, K), np.float32), "dur_lp": np.zeros((n_chunks, S, STEP_CAP, D), np.float32),
             "margin": np.zeros((n_chunks, S, STEP_CAP, 2), np.float32)}
        if rows:
            a["row_xor"] = np.zeros((n_chunks, S, STEP_CAP), np.uint32); a["row_sum"] = np.zeros((n_chunks, S, STEP_CAP), np.uint64)
        return a

    def record(a, i, s, r):
        n = r["n"]
        assert n <= STEP_CAP
        lp = r["label_lp"]
        order = np.argsort(-lp, axis=1, kind="stable")[:, :K].astype(np.int32)
        top = np.take_along_axis(lp, order, axis=1)
        dsort = np.sort(r["dur_lp"], axis=1)
        a["n"][i, s] = n
        a["labels"][i, s, :n], a["dur_idx"][i, s, :n] = r["labels"], r["dur_idx"]
        a["top_ids"][i, s, :n], a["top_lp"][i, s, :n], a["dur_lp"][i, s, :n] = order, top, r["dur_lp"]
        a["margin"][i, s, :n] = np.stack([top[:, 0] - top[:, 1], dsort[:, -1] - dsort[:, -2]], axis=1)
        if "row_xor" in a:
            u = np.ascontiguousarray(lp).view(np.uint32)
            a["row_xor"][i, s, :n] = np.bitwise_xor.reduce(u, axis=1)
            a["row_sum"][i, s, :n] = u.astype(np.uint64).sum(axis=1)
        assert np.array_equal(order[:, 0], r["labels"]), "a greedy path follows the first maximum"

    enc_n = np.zeros((n_chunks, S), np.int32)
    t0 = time.time()
    # ---- A: fp32 oracle, greedy --------------------------------------------------------------------------------------------------------
    om = oracle.Model(cfg, W)
    A = arrays("f32", True)
    enc_xor = np.zeros((n_chunks, S), np.uint32); enc_sum = np.zeros((n_chunks, S), np.uint64)
    streams = [oracle.Stream(om, LEFT, RIGHT) for _ in range(S)]
    shadow = [oracle.Stream(om, LEFT, RIGHT) for _ in range(S)]       # the plain greedy decode next to it: the scored walk IS the decode
    for i in range(n_chunks):
        for s, o in enumerate(streams):
            m = o.mel(pcm[s, i * CHUNK:(i + 1) * CHUNK])
            if m.shape[0] == 0:
                continue
            e = o.encode(m)
            enc_n[i, s] = e.shape[0]
            if e.shape[0] == 0:
                continue
            u = e.view(np.uint32).ravel()
            enc_xor[i, s] = np.bitwise_xor.reduce(u); enc_sum[i, s] = u.astype(np.uint64).sum()
            r = o.score(e)
            record(A, i, s, r)
            g = shadow[s].decode(e)
            assert [int(k) for k in r["labels"] if k != cfg.blank_id] == g["ids"].tolist(), "orc_stream_score's greedy walk == orc_stream_decode"
        if i % 8 == 7:
            print(f"fp32 chunk {i + 1}/{n_chunks}: {time.time() - t0:.1f}s", flush=True)
    for o in streams + shadow:
        o.close()
    del om
    # ---- B: bf16-mode oracle, greedy; C: the same model teacher-forced along A (the encoder does not depend on the decisions: shared) ----
    om = oracle.Model(dataclasses.replace(cfg, gemm_bf16=True), W)
    B = arrays("b16", False)
    C_top = np.zeros((n_chunks, S, STEP_CAP, K), np.float32); C_dur = np.zeros((n_chunks, S, STEP_CAP, D), np.float32)
    streams = [oracle.Stream(om, LEFT, RIGHT) for _ in range(S)]
    forced = [oracle.Stream(om, LEFT, RIGHT) for _ in range(S)]       # decode state only (never encodes)
    for i in range(n_chunks):
        for s, o in enumerate(streams):
            m = o.mel(pcm[s, i * CHUNK:(i + 1) * CHUNK])
            if m.shape[0] == 0:
                continue
            e = o.encode(m)
            assert e.shape[0] == enc_n[i, s]
            if e.shape[0] == 0:
                continue
            record(B, i, s, o.score(e))
            n = int(A["n"][i, s])
            r = forced[s].score(e, A["labels"][i, s, :n], A["dur_idx"][i, s, :n])
            assert r["n"] == n
            C_top[i, s, :n] = np.take_along_axis(r["label_lp"], A["top_ids"][i, s, :n].astype(np.int64), axis=1)
            C_dur[i, s, :n] = r["dur_lp"]
        if i % 8 == 7:
            print(f"bf16 chunk {i + 1}/{n_chunks}: {time.time() - t0:.1f}s", flush=True)
    for k, v in A.items():
        out["f32_" + k] = v
    for k, v in B.items():
        out["b16_" + k] = v
    out.update(enc_n=enc_n, f32_enc_xor=enc_xor, f32_enc_sum=enc_sum, b16_on_f32_top_lp=C_top, b16_on_f32_dur_lp=C_dur,
               oracle_seconds=np.array(time.time() - t0), oracle_threads=threads)
    valid = np.arange(STEP_CAP)[None, None, :] < A["n"][:, :, None]
    dl = np.abs(C_top - A["top_lp"])[valid]; dd = np.abs(C_dur - A["dur_lp"])[valid]
    ntok_a = int(((A["labels"] >= 0) & (A["labels"] != cfg.blank_id)).sum()); ntok_b = int(((B["labels"] >= 0) & (B["labels"] != cfg.blank_id)).sum())
    print(f"steps: fp32 path {int(A['n'].sum())} ({ntok_a} tokens), bf16 path {int(B['n'].sum())} ({ntok_b} tokens)")
    print(f"bf16-mode oracle along the fp32 path: label |dlogp| max {dl.max():.3e} mean {dl.mean():.3e}; duration max {dd.max():.3e} mean {dd.mean():.3e}")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
