"""Seeded synthetic inputs (no model weights, vocab or audio exist offline -- SURVEY.md fact 4).

* weights: the exact safetensors tensor names / shapes the reference loads
  (scripts/convert_nemo.py:98-310,409-446), fp32.
* vocab:   one SentencePiece piece per line (src/vocab.cpp:10-27).
* pcm:     16 kHz mono float32 noise bursts + sine sweeps in +-0.3.
"""
import os

import numpy as np

from .config import ModelConfig


def synth_weights(cfg: ModelConfig, seed: int = 42) -> dict:
    rng = np.random.default_rng(seed)
    W = {}

    def lin(name, out_f, in_f, bias=True, extra_shape=()):
        W[name + ".weight"] = (rng.standard_normal((out_f, in_f) + extra_shape) / np.sqrt(in_f)).astype(np.float32)
        if bias:
            W[name + ".bias"] = (0.02 * rng.standard_normal(out_f)).astype(np.float32)

    def norm(name, d):
        W[name + ".weight"] = (1.0 + 0.02 * rng.standard_normal(d)).astype(np.float32)
        W[name + ".bias"] = (0.02 * rng.standard_normal(d)).astype(np.float32)

    C, d, F = cfg.subsampling_channels, cfg.hidden_size, cfg.mel_bins
    f3 = ((((F - 1) // 2 + 1) - 1) // 2 + 1 - 1) // 2 + 1
    ep = getattr(cfg, "encoder_prefix", "encoder_.")
    p = ep + "subsampling_."
    for nm in ("conv1_", "dw1_", "dw2_"):
        W[p + nm + ".weight"] = (rng.standard_normal((C, 1, 3, 3)) / 3.0).astype(np.float32)
        W[p + nm + ".bias"] = (0.02 * rng.standard_normal(C)).astype(np.float32)
    for nm in ("conv2_", "conv3_"):
        lin(p + nm, C, C, extra_shape=(1, 1))
    lin(p + "proj_", d, C * f3)
    for i in range(cfg.num_layers):
        q = f"{ep}layers_.{i}."
        for ff in ("ffn1_", "ffn2_"):
            norm(q + ff + ".norm_", d)
            lin(q + ff + ".fc1_", cfg.ffn_intermediate, d)
            lin(q + ff + ".fc2_", d, cfg.ffn_intermediate)
        norm(q + "attn_.norm_", d)
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(q + "attn_.mha_." + nm, d, d)
        # keep the random encoder "alive": with unit-variance scores 17 softmax-averaging layers collapse the
        # time axis (every frame -> the utterance mean); sharper queries + a smaller attention branch keep
        # frame-to-frame structure, so decodes are non-degenerate and softmax sees a wide dynamic range.
        W[q + "attn_.mha_.q_proj.weight"] *= np.float32(4.0)
        W[q + "attn_.mha_.out_proj.weight"] *= np.float32(0.3)
        lin(q + "attn_.pos_proj_", d, d, bias=False)
        W[q + "attn_.pos_bias_u_"] = (0.1 * rng.standard_normal((cfg.num_heads, cfg.head_dim))).astype(np.float32)
        W[q + "attn_.pos_bias_v_"] = (0.1 * rng.standard_normal((cfg.num_heads, cfg.head_dim))).astype(np.float32)
        norm(q + "conv_.norm_", d)
        lin(q + "conv_.pointwise_conv1_", 2 * d, d, extra_shape=(1,))
        W[q + "conv_.depthwise_conv_.weight"] = (rng.standard_normal((d, 1, cfg.conv_kernel_size)) / 3.0).astype(np.float32)
        W[q + "conv_.depthwise_conv_.bias"] = (0.02 * rng.standard_normal(d)).astype(np.float32)
        norm(q + "conv_.batch_norm_", d)
        W[q + "conv_.batch_norm_.running_mean"] = (0.05 * rng.standard_normal(d)).astype(np.float32)
        W[q + "conv_.batch_norm_.running_var"] = rng.uniform(0.5, 1.5, d).astype(np.float32)
        W[q + "conv_.batch_norm_.num_batches_tracked"] = np.zeros((), np.float32)
        lin(q + "conv_.pointwise_conv2_", d, d, extra_shape=(1,))
        norm(q + "final_norm_", d)
    V, Hp, J = cfg.vocab_size, cfg.pred_hidden, cfg.joint_hidden
    if V <= 0:                                  # encoder-only (Sortformer's NEST encoder)
        return W
    emb = rng.standard_normal((V, Hp)).astype(np.float32)
    emb[cfg.blank_id if cfg.blank_id < V else V - 1] = 0.0     # blank/SOS row is zero by training (tdt.cpp:56-57)
    W["prediction_.embed_.weight"] = emb
    for l in range(cfg.num_lstm_layers):
        lin(f"prediction_.lstm_.cells_.{l}.input_proj_", 4 * Hp, Hp)      # bias = b_ih + b_hh (convert_nemo.py:409-417)
        lin(f"prediction_.lstm_.cells_.{l}.hidden_proj_", 4 * Hp, Hp, bias=False)
    jp = cfg.joint_prefix
    lin(jp + "enc_proj_", J, d)
    lin(jp + "pred_proj_", J, Hp)       # .bias is present in converted files but dropped by the reference (A5)
    if cfg.head == "rnnt":
        lin(jp + "out_proj_", V, J)
        W[jp + "out_proj_.bias"][cfg.blank_id] += 2.4
    else:
        lin(jp + "label_proj_", V, J)
        lin(jp + "duration_proj_", len(cfg.durations), J)
        # speech-like decode statistics on random weights: favour blank, and durations 1-2
        W[jp + "label_proj_.bias"][cfg.blank_id] += 2.4
        W[jp + "duration_proj_.bias"] += np.array([-1.0, 1.5, 0.5, -0.3, -0.8], np.float32)[: len(cfg.durations)]
    if cfg.ctc_vocab_size:
        lin("ctc_decoder_.proj_", cfg.ctc_vocab_size, d, extra_shape=(1,))
        W["ctc_decoder_.proj_.bias"][cfg.ctc_vocab_size - 1] += 3.5
    return W


def synth_sortformer_weights(sf, seed: int = 42) -> dict:
    """Sortformer state dict (AX_REGISTER_MODULES order, src/sortformer.cpp:41-47): nest_encoder_.*, projection_, transformer_.*,
    output_proj_, first_hidden_, hidden_to_spks_ (registered, unused by forward)."""
    W = synth_weights(sf.nest_encoder, seed)
    rng = np.random.default_rng(seed + 1000)
    d, dt, S, ffn = sf.nest_encoder.hidden_size, sf.transformer_hidden, sf.max_speakers, sf.transformer_ffn
    lin = lambda o, i: (rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32)
    vec = lambda n, s=0.02: (s * rng.standard_normal(n)).astype(np.float32)
    W["projection_.weight"], W["projection_.bias"] = lin(dt, d), vec(dt)
    for l in range(sf.transformer_layers):
        p = f"transformer_.layers_.{l}."
        for n in ("norm1_", "norm2_"):
            W[p + n + ".weight"] = (1 + vec(dt)).astype(np.float32); W[p + n + ".bias"] = vec(dt)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + "mha_." + n + ".weight"] = lin(dt, dt); W[p + "mha_." + n + ".bias"] = vec(dt)
        W[p + "mha_.q_proj.weight"] *= np.float32(3.0)               # sharper attention: keeps frame-to-frame structure alive
        W[p + "fc1_.weight"] = lin(ffn, dt); W[p + "fc1_.bias"] = vec(ffn)
        W[p + "fc2_.weight"] = lin(dt, ffn); W[p + "fc2_.bias"] = vec(dt)
    if sf.has_final_norm:
        W["transformer_.final_norm_.weight"] = (1 + vec(dt)).astype(np.float32); W["transformer_.final_norm_.bias"] = vec(dt)
    W["first_hidden_.weight"], W["first_hidden_.bias"] = lin(dt, dt), vec(dt)
    W["output_proj_.weight"], W["output_proj_.bias"] = (4.0 * lin(S, dt)).astype(np.float32), vec(S)   # activities on both sides of 0.5
    W["hidden_to_spks_.weight"], W["hidden_to_spks_.bias"] = lin(S, 2 * dt), vec(S)
    return W


def save_weights(path: str, W: dict):
    from safetensors.numpy import save_file
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    save_file({k: np.ascontiguousarray(v) for k, v in W.items()}, path)


def synth_vocab(n_pieces: int, seed: int = 7) -> list:
    rng = np.random.default_rng(seed)
    syll = ["ka", "to", "mi", "re", "su", "lo", "an", "ve", "di", "po", "qu", "ex", "ly", "ing", "er", "s", "t", "a", "o", "e"]
    pieces, seen = [], set()
    while len(pieces) < n_pieces:
        k = int(rng.integers(1, 4))
        w = "".join(syll[int(i)] for i in rng.integers(0, len(syll), k))
        w = ("▁" + w) if rng.random() < 0.55 else w
        if w not in seen:
            seen.add(w)
            pieces.append(w)
    return pieces


def save_vocab(path: str, pieces: list, with_scores: bool = True):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        for i, p in enumerate(pieces):
            f.write(f"{p}\t{-float(i):.4f}\n" if with_scores else p + "\n")


def synth_pcm(n_clips: int, n_samples: int, seed: int = 1234, sr: int = 16000) -> np.ndarray:
    """[n_clips, n_samples] float32 in +-0.3: noise bursts + sine sweeps (non-degenerate mel bins)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.empty((n_clips, n_samples), np.float32)
    for c in range(n_clips):
        x = 0.02 * rng.standard_normal(n_samples)
        for _ in range(6):
            f0, f1 = rng.uniform(80, 3500, 2)
            a = rng.uniform(0.02, 0.12)
            ph = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / max(t[-1], 1e-3))
            x += a * np.sin(ph + rng.uniform(0, 6.28))
        for _ in range(max(1, n_samples // 16000)):
            s = int(rng.integers(0, max(1, n_samples - 4000)))
            L = int(rng.integers(800, 4000))
            x[s:s + L] += 0.1 * rng.standard_normal(min(L, n_samples - s)) * np.hanning(min(L, n_samples - s))
        x = 0.3 * x / max(1e-9, np.max(np.abs(x)))
        out[c] = x.astype(np.float32)
    return out


def write_wav_pcm16(path: str, pcm: np.ndarray, sr: int = 16000):
    import wave
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(pcm, -1, 1) * 32767.0).astype("<i2").tobytes())
