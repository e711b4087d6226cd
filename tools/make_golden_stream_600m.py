#!/usr/bin/env python3
"""Generate tests/golden/nemotron600m_stream_depth24_seed42.npz: the FULL-DEPTH parity fixture of BASELINE configs[4] (nemotron-600m
streaming: d = 1024, 24 layers, 8 heads of 128, vocab 8193, 2 LSTM layers; att_context 70 / 1; 160 ms chunks).

Round-3 verdict: the GPU streaming tests stop at tiny and 2-layer cuts while every streaming product kernel was rewritten and configs[4] is
timed at a depth that is never parity-checked.  The 24-layer CPU oracle streams ~2.4 GB of fp32 weights per stream and chunk, too slow for
the GPU box's test run, so it is run ONCE here, in the authoring container:

  streams  = N_STREAMS sessions, each fed N_CHUNKS chunks of 2560 samples of synth_pcm(N_STREAMS, ..., seed=PCM_SEED)
  weights  = synth_weights(make_nemotron_600m_config(), seed=42)     (what tools/bench_stream.py and bench.py's also[] load)
  per chunk and stream (oracle.Stream = StreamingAudioPreprocessor::process_chunk, src/audio.cpp:195-259 ->
  StreamingFastConformerEncoder::forward_chunk, src/streaming_encoder.cpp:430-472 with cached attention :162-272 -> rnnt_streaming_decode_chunk,
  src/eou.cpp:17-98 as NemotronTranscriber::transcribe_chunk calls it, src/nemotron.cpp:24-52):
      mel_n / enc_n            frames produced (0 while audio is buffered)
      mel_bits / enc_bits      sum and xor of the uint32 bit patterns of the chunk's log-mel / encoder output
      enc_row0                 the first encoder row of the chunk (a readable sample next to the checksums)
      ids / start / end / conf_bits, n_tok      the tokens the chunk emitted
  ref_*    the same through the REFERENCE's own object code (oracle/_ref/libpk_ref_model.so: the streaming classes compiled where they lie) on
           stream 0: token ids and frames per chunk, and whether they equal the oracle's.

tests/test_gpu_stream.py::test_full_depth_nemotron_600m replays the audio through pk_stream_push / pk_stream_{mel,encode,decode} at 16
lock-step streams and requires every chunk's bits.  usage (authoring container, a few minutes): python tools/make_golden_stream_600m.py
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
OUT = os.path.join(ROOT, "tests", "golden", "nemotron600m_stream_depth24_seed42.npz")
N_STREAMS, N_CHUNKS, CHUNK, PCM_SEED, LEFT, RIGHT, MAX_TOK = 2, 40, 2560, 4242, 70, 1, 24


def bits_sum_xor(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()
    if u.size == 0:
        return np.zeros(2, np.uint64)
    return np.array([int(u.astype(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF), int(np.bitwise_xor.reduce(u))], np.uint64)


def main():
    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import synth
    import oracle
    import refmodel
    threads = min(8, os.cpu_count() or 1)
    oracle.set_threads(threads)
    cfg = pk.make_nemotron_600m_config()
    W = synth.synth_weights(cfg, seed=42)
    pcm = synth.synth_pcm(N_STREAMS, CHUNK * N_CHUNKS, seed=PCM_SEED)
    om = oracle.Model(cfg, W)
    S, d = N_STREAMS, cfg.hidden_size
    out = {"n_streams": S, "n_chunks": N_CHUNKS, "chunk": CHUNK, "pcm_seed": PCM_SEED, "weights_seed": 42, "att_left": LEFT, "att_right": RIGHT,
           "max_tok": MAX_TOK, "pcm_digest": np.asarray(pcm, np.float64).sum(axis=1)}
    mel_n = np.zeros((N_CHUNKS, S), np.int32); enc_n = np.zeros((N_CHUNKS, S), np.int32); n_tok = np.zeros((N_CHUNKS, S), np.int32)
    mel_bits = np.zeros((N_CHUNKS, S, 2), np.uint64); enc_bits = np.zeros((N_CHUNKS, S, 2), np.uint64)
    enc_row0 = np.zeros((N_CHUNKS, S, d), np.float32)
    ids = np.full((N_CHUNKS, S, MAX_TOK), -1, np.int32); st = np.zeros((N_CHUNKS, S, MAX_TOK), np.int32); en = np.zeros((N_CHUNKS, S, MAX_TOK), np.int32)
    cfb = np.zeros((N_CHUNKS, S, MAX_TOK), np.uint32)
    t0 = time.time()
    streams = [oracle.Stream(om, LEFT, RIGHT) for _ in range(S)]
    for i in range(N_CHUNKS):
        for s, o in enumerate(streams):
            m = o.mel(pcm[s, i * CHUNK:(i + 1) * CHUNK])
            mel_n[i, s] = m.shape[0]; mel_bits[i, s] = bits_sum_xor(m)
            if m.shape[0] == 0:
                continue
            e = o.encode(m)
            enc_n[i, s] = e.shape[0]; enc_bits[i, s] = bits_sum_xor(e)
            if e.shape[0] == 0:
                continue
            enc_row0[i, s] = e[0]
            r = o.decode(e)
            n = len(r["ids"])
            assert n <= MAX_TOK
            n_tok[i, s] = n
            ids[i, s, :n], st[i, s, :n], en[i, s, :n] = r["ids"], r["start"], r["end"]
            cfb[i, s, :n] = np.ascontiguousarray(r["conf"]).view(np.uint32)
        if i % 8 == 7:
            print(f"chunk {i + 1}/{N_CHUNKS}: {time.time() - t0:.1f}s, tokens so far {n_tok.sum(axis=0).tolist()}", flush=True)
    out.update(mel_n=mel_n, enc_n=enc_n, n_tok=n_tok, mel_bits=mel_bits, enc_bits=enc_bits, enc_row0=enc_row0, ids=ids, start=st, end=en, conf_bits=cfb,
               oracle_seconds=np.array(time.time() - t0), oracle_threads=threads)
    print(f"oracle: {N_CHUNKS} chunks x {S} streams in {time.time() - t0:.1f}s; tokens per stream {n_tok.sum(axis=0).tolist()}", flush=True)
    del streams

    if refmodel.available():
        t1 = time.time()
        with tempfile.TemporaryDirectory() as td:
            wp = os.path.join(td, "w.safetensors")
            synth.save_weights(wp, W)
            rm = refmodel.Model(cfg, wp, kind="nemotron", att_left=LEFT, att_right=RIGHT)
            sr = refmodel.Stream(rm)
            r_ids = np.full((N_CHUNKS, MAX_TOK), -1, np.int32); r_n = np.zeros(N_CHUNKS, np.int32)
            r_st = np.zeros((N_CHUNKS, MAX_TOK), np.int32)
            for i in range(N_CHUNKS):
                m = sr.mel(pcm[0, i * CHUNK:(i + 1) * CHUNK])
                if m.shape[0] == 0:
                    continue
                e = sr.encode(m)
                if e.shape[0] == 0:
                    continue
                a, b, _, _ = sr.decode(e, blank_id=cfg.blank_id, max_symbols=cfg.max_symbols_per_step)
                r_n[i] = len(a); r_ids[i, :len(a)] = a; r_st[i, :len(a)] = b
            del sr, rm
        same = bool(np.array_equal(r_n, n_tok[:, 0]) and np.array_equal(r_ids, ids[:, 0]) and
                    all(np.array_equal(r_st[i, :r_n[i]], st[i, 0, :r_n[i]]) for i in range(N_CHUNKS)))
        out.update(ref_ids=r_ids, ref_n=r_n, ref_start=r_st, ref_equal_oracle=np.array(same), ref_seconds=np.array(time.time() - t1))
        print(f"reference code (StreamingFastConformerEncoder + rnnt_streaming_decode_chunk) on stream 0: {time.time() - t1:.1f}s; ids / frames identical "
              f"to the oracle: {same}", flush=True)
    else:
        print("oracle/_ref absent: no reference-code ids in the fixture", flush=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
