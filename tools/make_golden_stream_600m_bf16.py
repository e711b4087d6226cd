#!/usr/bin/env python3
"""Generate tests/golden/nemotron600m_stream_bf16_depth24_seed42.npz: the full-depth fixture of BASELINE configs[4] in the TOLERANCE-class mode
(pk_config.gemm_bf16: every Linear / 1x1-conv product of the chunk on bf16 operands with fp32 accumulation -- kernels/gemm_smallm_bf16.hip).

Same sessions as tools/make_golden_stream_600m.py (nemotron-600m as shipped: 24 layers, d 1024, vocab 8193, 2 LSTM layers; att_context 70 / 1;
2 streams x 40 chunks of 2560 samples of synth_pcm(seed 4242); synth_weights(seed 42)) through the oracle's Stream with gemm_bf16 = 1 -- the
specification of the mode (oracle/pk_oracle.c: linear_t rounds both operands to bf16; the reference's loops are StreamingFastConformerEncoder::
forward_chunk, src/streaming_encoder.cpp:430-472 with cached attention :162-272 and causal conv :41-78, rnnt_streaming_decode_chunk,
src/eou.cpp:17-98).  The mode is compared within a tolerance, so the fixture holds VALUES, not checksums:
    enc[chunk][stream][frame][d]   the chunk's encoder output (fp32)          enc_n   frames per chunk
    ids / start / end / n_tok      the tokens each chunk emitted
    step_label / step_margin / n_steps    EVERY decision of each chunk's greedy loop, in order: the label chosen (blank included) and its top-1 /
                                   top-2 log-prob margin (label and duration heads) -- what the mode's token statement walks (oracle/tolerance.py)
    fp32_row0_gap                  |bf16-mode row 0 - fp32-mode row 0| per chunk and stream, max and mean (fp32 rows from the fp32 fixture): the
                                   size of the mode's own deviation from fp32, the yardstick the GPU's deviation from this fixture is held against
usage (authoring container, a few minutes of CPU): python tools/make_golden_stream_600m_bf16.py"""
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
OUT = os.path.join(ROOT, "tests", "golden", "nemotron600m_stream_bf16_depth24_seed42.npz")
FP32 = os.path.join(ROOT, "tests", "golden", "nemotron600m_stream_depth24_seed42.npz")
N_STREAMS, N_CHUNKS, CHUNK, PCM_SEED, LEFT, RIGHT, MAX_TOK = 2, 40, 2560, 4242, 70, 1, 24


def main():
    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import synth
    import oracle
    threads = min(8, os.cpu_count() or 1)
    oracle.set_threads(threads)
    cfg = dataclasses.replace(pk.make_nemotron_600m_config(), gemm_bf16=True)
    W = synth.synth_weights(cfg, seed=42)
    pcm = synth.synth_pcm(N_STREAMS, CHUNK * N_CHUNKS, seed=PCM_SEED)
    om = oracle.Model(cfg, W)
    S, d = N_STREAMS, cfg.hidden_size
    out = {"n_streams": S, "n_chunks": N_CHUNKS, "chunk": CHUNK, "pcm_seed": PCM_SEED, "weights_seed": 42, "att_left": LEFT, "att_right": RIGHT,
           "max_tok": MAX_TOK, "pcm_digest": np.asarray(pcm, np.float64).sum(axis=1)}
    enc_n = np.zeros((N_CHUNKS, S), np.int32); n_tok = np.zeros((N_CHUNKS, S), np.int32)
    enc = np.zeros((N_CHUNKS, S, 2, d), np.float32)
    ids = np.full((N_CHUNKS, S, MAX_TOK), -1, np.int32); st = np.zeros((N_CHUNKS, S, MAX_TOK), np.int32); en = np.zeros((N_CHUNKS, S, MAX_TOK), np.int32)
    STEP_CAP = 2 * (cfg.max_symbols_per_step + 1) + 16
    step_label = np.full((N_CHUNKS, S, STEP_CAP), -1, np.int32); step_margin = np.zeros((N_CHUNKS, S, STEP_CAP), np.float32)
    n_steps = np.zeros((N_CHUNKS, S), np.int32)
    t0 = time.time()
    streams = [oracle.Stream(om, LEFT, RIGHT) for _ in range(S)]
    for i in range(N_CHUNKS):
        for s, o in enumerate(streams):
            m = o.mel(pcm[s, i * CHUNK:(i + 1) * CHUNK])
            if m.shape[0] == 0:
                continue
            e = o.encode(m)
            enc_n[i, s] = e.shape[0]
            if e.shape[0] == 0:
                continue
            assert e.shape[0] <= 2
            enc[i, s, :e.shape[0]] = e
            r = o.decode(e, margins=True)
            n = len(r["ids"])
            assert n <= MAX_TOK
            n_tok[i, s] = n
            ids[i, s, :n], st[i, s, :n], en[i, s, :n] = r["ids"], r["start"], r["end"]
            k = len(r["step_label"])
            n_steps[i, s] = k
            step_label[i, s, :k], step_margin[i, s, :k] = r["step_label"], r["step_margin"]
        if i % 8 == 7:
            print(f"chunk {i + 1}/{N_CHUNKS}: {time.time() - t0:.1f}s, tokens so far {n_tok.sum(axis=0).tolist()}", flush=True)
    out.update(enc_n=enc_n, n_tok=n_tok, enc=enc, ids=ids, start=st, end=en, step_label=step_label, step_margin=step_margin, n_steps=n_steps, oracle_seconds=np.array(time.time() - t0), oracle_threads=threads)
    if os.path.exists(FP32):
        f = np.load(FP32, allow_pickle=False)
        assert np.array_equal(f["enc_n"], enc_n), "the fp32 fixture was generated on other sessions"
        gap = np.abs(enc[:, :, 0, :] - f["enc_row0"])
        have = enc_n > 0
        out.update(fp32_row0_gap_max=np.array(float(gap[have].max())), fp32_row0_gap_mean=np.array(float(gap[have].mean())),
                   fp32_row0_absmax=np.array(float(np.abs(f["enc_row0"][have]).max())),
                   fp32_tokens=np.array(int(f["n_tok"].sum())), fp32_ids_equal=np.array(bool(np.array_equal(f["ids"], ids))))
        print(f"bf16-mode vs fp32-mode oracle, first encoder row of every chunk: max {gap[have].max():.3e} mean {gap[have].mean():.3e} "
              f"(max|x| {np.abs(f['enc_row0'][have]).max():.2f}); token ids identical to the fp32 run: {bool(np.array_equal(f['ids'], ids))}")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(f"oracle (gemm_bf16): {N_CHUNKS} chunks x {S} streams in {time.time() - t0:.1f}s; tokens per stream {n_tok.sum(axis=0).tolist()}; "
          f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
