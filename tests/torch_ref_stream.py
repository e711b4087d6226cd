"""Independent torch-CPU restatement of the reference's streaming path, written from the reference sources with tensor ops
(cat / matmul / softmax / masked_fill) rather than the oracle's scalar loops: StreamingAudioPreprocessor::process_chunk
(src/audio.cpp:195-259), CausalConvSubsampling::forward_cached (src/streaming_encoder.cpp:348-385), StreamingConformerBlock::
forward_cached (:289-301) with StreamingConformerAttention::forward_cached (:162-272) and CausalConformerConvModule::
forward_cached (:41-78).  Cross-checks oracle/pk_oracle.c's streaming section (tests/test_stream_oracle.py)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


class TorchStream:
    def __init__(self, cfg, W, fb, att_left, att_right, bf16=False):
        """bf16: the tolerance-class mode of the streaming path -- every Linear / 1x1-conv product takes both operands rounded to bf16 (RNE), fp32
        accumulation; everything else (3x3 / depthwise convs, LayerNorm, attention arithmetic, caches) stays fp32 (oracle/pk_oracle.c, streaming section)."""
        self.bf16 = bf16
        self.cfg, self.W, self.L, self.R = cfg, {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in W.items()}, att_left, att_right
        self.fb = torch.from_numpy(fb)                      # [257][n_mels]
        self.last = 0.0
        self.overlap = torch.zeros(0)
        self.mel_cache = None
        self.caches = [dict(k=None, v=None, conv=None) for _ in range(cfg.num_layers)]

    def mel(self, pcm):
        x = torch.from_numpy(np.asarray(pcm, np.float32))
        prev = torch.cat([torch.tensor([self.last], dtype=torch.float32), x[:-1]])
        pre = x - 0.97 * prev
        self.last = float(x[-1])
        buf = torch.cat([self.overlap, pre])
        if buf.numel() < 400:
            self.overlap = buf
            return None
        n_frames = (buf.numel() - 400) // 160 + 1
        consumed = (n_frames - 1) * 160 + 400
        self.overlap = buf[consumed:].clone()
        frames = buf[:consumed].unfold(0, 400, 160)          # [n_frames][400]
        win = torch.hann_window(400, periodic=False, dtype=torch.float64).to(torch.float32)
        spec = torch.fft.rfft(F.pad(frames * win, (0, 112)), n=512)
        power = spec.abs() ** 2                              # [n_frames][257]
        return torch.log(power @ self.fb + 2.0 ** -24)       # [n_frames][n_mels]

    def w(self, name):
        return self.W[name]

    def rb(self, t):                                         # operand rounding of the bf16 mode
        return t.bfloat16().float() if self.bf16 else t

    def linear(self, v, w, b=None):                          # the mode's product: rounded operands, fp32 accumulate
        return F.linear(self.rb(v), self.rb(w), b)

    def subsample(self, mel):                                # ConvSubsampling::forward on one chunk (ReLU, src/encoder.cpp:219-241)
        p = "encoder_.subsampling_."
        C = self.cfg.subsampling_channels
        x = mel[None, None]
        x = F.relu(F.conv2d(x, self.w(p + "conv1_.weight"), self.w(p + "conv1_.bias"), stride=2, padding=1))
        x = F.conv2d(x, self.w(p + "dw1_.weight"), self.w(p + "dw1_.bias"), stride=2, padding=1, groups=C)
        x = F.relu(F.conv2d(self.rb(x), self.rb(self.w(p + "conv2_.weight")), self.w(p + "conv2_.bias")))
        x = F.conv2d(x, self.w(p + "dw2_.weight"), self.w(p + "dw2_.bias"), stride=2, padding=1, groups=C)
        x = F.relu(F.conv2d(self.rb(x), self.rb(self.w(p + "conv3_.weight")), self.w(p + "conv3_.bias")))
        b, c, t, f = x.shape
        x = x.permute(0, 2, 1, 3).reshape(b, t, c * f)
        return self.linear(x, self.w(p + "proj_.weight"), self.w(p + "proj_.bias"))[0]

    def pos_emb(self, T, d):
        pe = torch.zeros(2 * T - 1, d)
        pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]
        div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        return pe

    def encode(self, mel):
        if self.mel_cache is not None:
            mel = torch.cat([self.mel_cache, mel], 0)
        total = mel.shape[0]
        consumable = (total // 8) * 8
        self.mel_cache = mel[consumable:].clone() if total > consumable else None
        if consumable == 0:
            return None
        x = self.subsample(mel[:consumable])
        c, d = x.shape
        H = self.cfg.num_heads
        hd = d // H
        pe = self.pos_emb(self.L + c, d)
        for l, cache in enumerate(self.caches):
            q = f"encoder_.layers_.{l}."
            ln = lambda v, n: F.layer_norm(v, (d,), self.w(q + n + ".weight"), self.w(q + n + ".bias"), 1e-5)
            lin = lambda v, n, bias=True: self.linear(v, self.w(q + n + ".weight"), self.w(q + n + ".bias") if bias else None)
            ffn = lambda v, n: v + 0.5 * lin(F.silu(lin(ln(v, n + "norm_"), n + "fc1_")), n + "fc2_")
            x = ffn(x, "ffn1_.")
            # attention with K/V cache
            n_ = ln(x, "attn_.norm_")
            qq, kk, vv = (lin(n_, "attn_.mha_." + t).reshape(c, H, hd).transpose(0, 1) for t in ("q_proj", "k_proj", "v_proj"))
            if cache["k"] is not None:
                kk, vv = torch.cat([cache["k"], kk], 1), torch.cat([cache["v"], vv], 1)
            kv = kk.shape[1]
            cache["k"], cache["v"] = (kk[:, kv - self.L:], vv[:, kv - self.L:]) if kv > self.L else (kk, vv)
            u, v_ = self.w(q + "attn_.pos_bias_u_").reshape(H, 1, hd), self.w(q + "attn_.pos_bias_v_").reshape(H, 1, hd)
            content = (qq + u) @ kk.transpose(1, 2)
            p = lin(pe, "attn_.pos_proj_", bias=False).reshape(-1, H, hd).transpose(0, 1)
            pos = (qq + v_) @ p.transpose(1, 2)
            if pos.shape[2] > kv:
                pos = pos[:, :, pos.shape[2] - kv:]
            scores = (content + pos) * (1.0 / math.sqrt(hd))
            qi = torch.arange(c)[:, None] + (kv - c)
            dist = qi - torch.arange(kv)[None, :]
            scores = scores.masked_fill(((dist > self.L) | (-dist > self.R))[None], -1e9)
            out = (torch.softmax(scores, -1) @ vv).transpose(0, 1).reshape(c, d)
            x = x + lin(out, "attn_.mha_.out_proj")
            # causal conv module with cache
            g = F.glu(lin_conv(self, q + "conv_.pointwise_conv1_", ln(x, "conv_.norm_")), dim=-1).transpose(0, 1)   # [d][c]
            K = self.cfg.conv_kernel_size
            cat = torch.cat([cache["conv"] if cache["conv"] is not None else torch.zeros(d, K - 1), g], 1)
            cache["conv"] = cat[:, cat.shape[1] - (K - 1):].clone()
            y = F.conv1d(cat[None], self.w(q + "conv_.depthwise_conv_.weight"), self.w(q + "conv_.depthwise_conv_.bias"), groups=d)[0]
            y = F.batch_norm(y[None], self.w(q + "conv_.batch_norm_.running_mean"), self.w(q + "conv_.batch_norm_.running_var"),
                             self.w(q + "conv_.batch_norm_.weight"), self.w(q + "conv_.batch_norm_.bias"), False, 0.0, 1e-5)[0]
            y = lin_conv(self, q + "conv_.pointwise_conv2_", F.silu(y).transpose(0, 1))
            x = x + y
            x = ffn(x, "ffn2_.")
            x = ln(x, "final_norm_")
        return x


def lin_conv(ts, name, v):        # 1x1 Conv1d stored as [out][in][1]
    w = ts.w(name + ".weight")
    return ts.linear(v, w.reshape(w.shape[0], -1), ts.w(name + ".bias"))
