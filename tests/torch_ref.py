"""A second, independent restatement of the reference hot path in torch-CPU ops
(F.conv2d, F.layer_norm, torch.stft, ...), modelled on the reference author's own
scripts/compare_encoder.py:24-284 and scripts/compare_features.py:22-57.  It exists only
to catch transcription errors in the C oracle (tests/test_oracle_vs_torch.py): the two
agree to fp32 round-off, not bit-for-bit (torch picks its own summation orders)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def mel_features(pcm, fb, n_mels=80, window_centered=True):
    x = t(pcm).to(torch.float32)
    x = torch.cat([x[:1], x[1:] - 0.97 * x[:-1]])                       # src/audio.cpp:104-114
    win = torch.hann_window(400, periodic=False, dtype=torch.float64).to(torch.float32)
    w = torch.zeros(512)
    off = 56 if window_centered else 0
    w[off:off + 400] = win
    st = torch.stft(x, n_fft=512, hop_length=160, win_length=512, window=w, center=True,
                    pad_mode="reflect", return_complex=True)            # :117-120
    power = st.abs() ** 2                                               # :123-124
    mel = t(fb).T @ power                                               # :132
    logmel = torch.log(mel + 2.0 ** -24)                                # :135-136
    mean = logmel.mean(dim=1, keepdim=True)
    std = logmel.std(dim=1, keepdim=True, correction=1)                 # :140-149
    return ((logmel - mean) / (std + 1e-5)).T.contiguous().numpy(), logmel.numpy()


def subsampling(W, feats):
    p = "encoder_.subsampling_."
    x = t(feats).unsqueeze(1)
    C = W[p + "conv1_.weight"].shape[0]
    x = F.relu(F.conv2d(x, t(W[p + "conv1_.weight"]), t(W[p + "conv1_.bias"]), stride=2, padding=1))
    x = F.conv2d(x, t(W[p + "dw1_.weight"]), t(W[p + "dw1_.bias"]), stride=2, padding=1, groups=C)
    x = F.relu(F.conv2d(x, t(W[p + "conv2_.weight"]), t(W[p + "conv2_.bias"])))
    x = F.conv2d(x, t(W[p + "dw2_.weight"]), t(W[p + "dw2_.bias"]), stride=2, padding=1, groups=C)
    x = F.relu(F.conv2d(x, t(W[p + "conv3_.weight"]), t(W[p + "conv3_.bias"])))
    b, c, tt, f = x.shape
    x = x.permute(0, 2, 1, 3).reshape(b, tt, c * f)
    return F.linear(x, t(W[p + "proj_.weight"]), t(W[p + "proj_.bias"])).numpy()


def rel_shift(x):                                                       # src/encoder.cpp:85-109, literally
    b, h, n, p = x.shape
    x = F.pad(x, (1, 0))
    x = x.reshape(b, h, p + 1, n)[:, :, 1:, :].reshape(b, h, n, p)
    return x[..., :n]


def conformer_block(W, layer, x, pe, n_heads, stop_after=0, bf16=False):
    """bf16: the tolerance-class mode (pk_config.gemm_bf16) stated independently of oracle/pk_oracle.c -- every Linear / 1x1-conv product takes
    both operands rounded to bf16 (torch's RNE) with fp32 accumulation; for head sizes 64 / 128 the attention is the mode's bf16 form: q, k, v
    and the projected position table stored as bf16, ONE biased query copy qu = bf16(q + u) for both score terms with the position term completed
    by c[p] = (v - u) . P_p in fp32, probabilities rounded to bf16 for the value product, the normaliser not, the context stored as bf16."""
    q = f"encoder_.layers_.{layer}."
    x = t(x)
    d = x.shape[-1]
    rb = (lambda v: v.bfloat16().float()) if bf16 else (lambda v: v)

    def lin(v, w, b=None):
        return F.linear(rb(v), rb(w), b)

    def ln(name, v):
        return F.layer_norm(v, (d,), t(W[q + name + ".weight"]), t(W[q + name + ".bias"]), 1e-5)

    def ffn(name, v):
        h = lin(ln(name + ".norm_", v), t(W[q + name + ".fc1_.weight"]), t(W[q + name + ".fc1_.bias"]))
        h = lin(F.silu(h), t(W[q + name + ".fc2_.weight"]), t(W[q + name + ".fc2_.bias"]))
        return v + 0.5 * h

    x = ffn("ffn1_", x)
    if stop_after == 1:
        return x.numpy()
    # attention  src/encoder.cpp:111-186
    n = ln("attn_.norm_", x)
    B, T, _ = n.shape
    hd = d // n_heads
    qq = lin(n, t(W[q + "attn_.mha_.q_proj.weight"]), t(W[q + "attn_.mha_.q_proj.bias"])).view(B, T, n_heads, hd).transpose(1, 2)
    kk = lin(n, t(W[q + "attn_.mha_.k_proj.weight"]), t(W[q + "attn_.mha_.k_proj.bias"])).view(B, T, n_heads, hd).transpose(1, 2)
    vv = lin(n, t(W[q + "attn_.mha_.v_proj.weight"]), t(W[q + "attn_.mha_.v_proj.bias"])).view(B, T, n_heads, hd).transpose(1, 2)
    u = t(W[q + "attn_.pos_bias_u_"]).view(1, n_heads, 1, hd)
    v_ = t(W[q + "attn_.pos_bias_v_"]).view(1, n_heads, 1, hd)
    p = lin(t(pe), t(W[q + "attn_.pos_proj_.weight"])).view(1, -1, n_heads, hd).transpose(1, 2)
    if bf16 and hd in (64, 128):
        qq, kk, vv, p = rb(qq), rb(kk), rb(vv), rb(p)
        qu = rb(qq + u)
        content = qu @ kk.transpose(-1, -2)
        cvec = ((v_ - u) @ p.transpose(-1, -2))                                   # [1][H][1][P]
        pos = rel_shift(qu @ p.transpose(-1, -2) + cvec)
        sc = (content + pos) * (1.0 / math.sqrt(hd))
        e = torch.exp(sc - sc.max(dim=-1, keepdim=True).values)
        o = rb((rb(e) @ vv) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, T, d)
    else:
        content = (qq + u) @ kk.transpose(-1, -2)
        pos = rel_shift((qq + v_) @ p.transpose(-1, -2))
        a = torch.softmax((content + pos) * (1.0 / math.sqrt(hd)), dim=-1)
        o = (a @ vv).transpose(1, 2).reshape(B, T, d)
    x = x + lin(o, t(W[q + "attn_.mha_.out_proj.weight"]), t(W[q + "attn_.mha_.out_proj.bias"]))
    if stop_after == 2:
        return x.numpy()
    # conv module  src/encoder.cpp:59-75
    n = ln("conv_.norm_", x).transpose(1, 2)
    y = F.glu(F.conv1d(rb(n), rb(t(W[q + "conv_.pointwise_conv1_.weight"])), t(W[q + "conv_.pointwise_conv1_.bias"])), dim=1)
    K = W[q + "conv_.depthwise_conv_.weight"].shape[-1]
    y = F.conv1d(y, t(W[q + "conv_.depthwise_conv_.weight"]), t(W[q + "conv_.depthwise_conv_.bias"]), padding=(K - 1) // 2, groups=d)
    y = F.batch_norm(y, t(W[q + "conv_.batch_norm_.running_mean"]), t(W[q + "conv_.batch_norm_.running_var"]),
                     t(W[q + "conv_.batch_norm_.weight"]), t(W[q + "conv_.batch_norm_.bias"]), training=False, eps=1e-5)
    y = F.conv1d(rb(F.silu(y)), rb(t(W[q + "conv_.pointwise_conv2_.weight"])), t(W[q + "conv_.pointwise_conv2_.bias"]))
    x = x + y.transpose(1, 2)
    if stop_after == 3:
        return x.numpy()
    x = ffn("ffn2_", x)
    if stop_after == 4:
        return x.numpy()
    return ln("final_norm_", x).numpy()


def pos_emb(T, d):                                                      # src/encoder.cpp:9-30 (float math)
    pe = np.zeros((2 * T - 1, d), np.float32)
    for p in range(2 * T - 1):
        pos = np.float32(T - 1 - p)
        i = np.arange(0, d, 2, dtype=np.float32)
        div = np.exp(i * np.float32(-np.log(np.float32(10000.0)) / np.float32(d))).astype(np.float32)
        pe[p, 0::2] = np.sin(pos * div)
        pe[p, 1::2] = np.cos(pos * div)
    return pe


def encoder(W, cfg, feats):
    x = subsampling(W, feats)
    pe = pos_emb(x.shape[1], x.shape[2])
    for l in range(cfg.num_layers):
        x = conformer_block(W, l, x, pe, cfg.num_heads)
    return x


def ctc_logprobs(W, enc):
    w = t(W["ctc_decoder_.proj_.weight"]).squeeze(-1)
    return torch.log_softmax(F.linear(t(enc), w, t(W["ctc_decoder_.proj_.bias"])), dim=-1).numpy()


def tdt_greedy(W, cfg, enc, max_steps=100000):
    """src/tdt.cpp:36-110 with src/rnnt.cpp:22-28 and src/lstm.cpp:11-49, one utterance at a time."""
    jp = cfg.joint_prefix
    E = t(W["prediction_.embed_.weight"])
    L, Hp = cfg.num_lstm_layers, cfg.pred_hidden
    out = []
    for b in range(enc.shape[0]):
        e = t(enc[b])
        ep = F.linear(e, t(W[jp + "enc_proj_.weight"]), t(W[jp + "enc_proj_.bias"]))
        h = [torch.zeros(Hp) for _ in range(L)]
        c = [torch.zeros(Hp) for _ in range(L)]
        tok, tt, ids, steps = cfg.blank_id, 0, [], 0
        T = e.shape[0]
        while tt < T and steps < max_steps:
            for _ in range(cfg.max_symbols_per_step):
                sh, sc = [v.clone() for v in h], [v.clone() for v in c]
                x = E[tok]
                for l in range(L):
                    g = F.linear(x, t(W[f"prediction_.lstm_.cells_.{l}.input_proj_.weight"]), t(W[f"prediction_.lstm_.cells_.{l}.input_proj_.bias"])) \
                        + F.linear(h[l], t(W[f"prediction_.lstm_.cells_.{l}.hidden_proj_.weight"]))
                    i, f, gg, o = g.chunk(4)
                    c[l] = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
                    h[l] = torch.sigmoid(o) * torch.tanh(c[l])
                    x = h[l]
                z = F.relu(ep[tt] + F.linear(x, t(W[jp + "pred_proj_.weight"])))
                lab = torch.log_softmax(F.linear(z, t(W[jp + "label_proj_.weight"]), t(W[jp + "label_proj_.bias"])), -1)
                dur = torch.log_softmax(F.linear(z, t(W[jp + "duration_proj_.weight"]), t(W[jp + "duration_proj_.bias"])), -1)
                steps += 1
                k, di = int(lab.argmax()), int(dur.argmax())
                skip = cfg.durations[di] if di < len(cfg.durations) else 1
                if k == cfg.blank_id:
                    h, c = sh, sc
                    tt += max(skip, 1)
                    break
                ids.append(k)
                tok = k
                if skip > 0:
                    tt += skip
                    break
        out.append(ids)
    return out


def tdt_score(W, cfg, enc, labels, dur_idx, bf16=False):
    """The TDT loop (src/tdt.cpp:62-106) of ONE utterance enc[T][d] along a GIVEN decision path, returning every step's label and duration
    log-probs.  bf16: the tolerance-class mode's decode rules stated independently of oracle/pk_oracle.c -- enc_proj with both operands rounded
    to bf16, the hidden-to-hidden and upper-layer input projections, pred_proj and both heads with bf16 WEIGHTS, h' = bf16(o tanh(c')) and
    z = bf16(relu(enc_proj + pred_proj)) stored rounded; the layer-0 input projection (an embedding row times W_ih) and the cell state stay fp32."""
    rb = (lambda v: v.bfloat16().float()) if bf16 else (lambda v: v)
    jp = cfg.joint_prefix
    E = t(W["prediction_.embed_.weight"])
    L, Hp = cfg.num_lstm_layers, cfg.pred_hidden
    e = t(enc)
    ep = F.linear(rb(e), rb(t(W[jp + "enc_proj_.weight"])), t(W[jp + "enc_proj_.bias"]))
    h = [torch.zeros(Hp) for _ in range(L)]
    c = [torch.zeros(Hp) for _ in range(L)]
    tok, tt, T = cfg.blank_id, 0, e.shape[0]
    lab_rows, dur_rows = [], []
    for k, di in zip(labels, dur_idx):
        if tt >= T:
            break
        sh, sc = [v.clone() for v in h], [v.clone() for v in c]
        x = E[tok]
        for l in range(L):
            wih = t(W[f"prediction_.lstm_.cells_.{l}.input_proj_.weight"])
            g = F.linear(x, rb(wih) if l > 0 else wih, t(W[f"prediction_.lstm_.cells_.{l}.input_proj_.bias"])) \
                + F.linear(h[l], rb(t(W[f"prediction_.lstm_.cells_.{l}.hidden_proj_.weight"])))
            i, f, gg, o = g.chunk(4)
            c[l] = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
            h[l] = rb(torch.sigmoid(o) * torch.tanh(c[l]))
            x = h[l]
        z = rb(F.relu(ep[tt] + F.linear(x, rb(t(W[jp + "pred_proj_.weight"])))))
        lab_rows.append(torch.log_softmax(F.linear(z, rb(t(W[jp + "label_proj_.weight"])), t(W[jp + "label_proj_.bias"])), -1).numpy())
        dur_rows.append(torch.log_softmax(F.linear(z, rb(t(W[jp + "duration_proj_.weight"])), t(W[jp + "duration_proj_.bias"])), -1).numpy())
        skip = cfg.durations[int(di)] if int(di) < len(cfg.durations) else 1
        if int(k) == cfg.blank_id:
            h, c = sh, sc
            tt += max(skip, 1)
        else:
            tok = int(k)
            tt += skip
    return np.stack(lab_rows), np.stack(dur_rows)
