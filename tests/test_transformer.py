"""TransformerEncoder / TransformerBlock (reference src/transformer.cpp:15-88; north_star names the file, Sortformer is its only
user in the reference).  CPU: the oracle's restatement against an independent torch restatement (fp32 round-off).  GPU: the
product (pk_transformer_*) against the oracle, bit for bit -- including heads of 24 features (Sortformer's 192 / 8), which the
product zero-pads to the 32-wide MFMA k-block."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from parakeet_cpp_amd import synth


def make_weights(prefix, d, L, ffn, final_norm, seed):
    rng = np.random.default_rng(seed)
    W = {}
    lin = lambda o, i: (rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32)
    vec = lambda n, s=0.02: (s * rng.standard_normal(n)).astype(np.float32)
    for l in range(L):
        p = f"{prefix}layers_.{l}."
        for n in ("norm1_", "norm2_"):
            W[p + n + ".weight"] = (1 + vec(d)).astype(np.float32); W[p + n + ".bias"] = vec(d)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + "mha_." + n + ".weight"] = lin(d, d); W[p + "mha_." + n + ".bias"] = vec(d)
        W[p + "fc1_.weight"] = lin(ffn, d); W[p + "fc1_.bias"] = vec(ffn)
        W[p + "fc2_.weight"] = lin(d, ffn); W[p + "fc2_.bias"] = vec(d)
    if final_norm:
        W[prefix + "final_norm_.weight"] = (1 + vec(d)).astype(np.float32); W[prefix + "final_norm_.bias"] = vec(d)
    return W


def torch_transformer(W, prefix, x, L, H, pre_ln, final_norm):
    t = lambda k: torch.from_numpy(W[prefix + k])
    x = torch.from_numpy(x)
    d = x.shape[-1]
    hd = d // H
    for l in range(L):
        p = f"layers_.{l}."
        ln = lambda v, n: F.layer_norm(v, (d,), t(p + n + ".weight"), t(p + n + ".bias"), 1e-5)
        a_in = ln(x, "norm1_") if pre_ln else x
        B, T, _ = x.shape
        q, k, v = (F.linear(a_in, t(p + f"mha_.{n}.weight"), t(p + f"mha_.{n}.bias")).reshape(B, T, H, hd).transpose(1, 2)
                   for n in ("q_proj", "k_proj", "v_proj"))
        att = torch.softmax(q @ k.transpose(-1, -2) * (1.0 / hd ** 0.5), -1) @ v
        out = F.linear(att.transpose(1, 2).reshape(B, T, d), t(p + "mha_.out_proj.weight"), t(p + "mha_.out_proj.bias"))
        x = x + out if pre_ln else ln(x + out, "norm1_")
        f_in = ln(x, "norm2_") if pre_ln else x
        y = F.linear(F.relu(F.linear(f_in, t(p + "fc1_.weight"), t(p + "fc1_.bias"))), t(p + "fc2_.weight"), t(p + "fc2_.bias"))
        x = x + y if pre_ln else ln(x + y, "norm2_")
    if final_norm:
        x = F.layer_norm(x, (d,), t("final_norm_.weight"), t("final_norm_.bias"), 1e-5)
    return x.numpy()


CASES = [(192, 1, 8, 768, False, False, 1300),      # longer than the LDS score block holds: global-scratch attention
         (192, 3, 8, 768, True, False, 37), (192, 2, 8, 768, False, True, 50), (128, 2, 2, 256, True, True, 201), (256, 1, 8, 512, True, False, 9)]


def oracle_model(orc, W):
    from conftest import pk
    return orc.Model(pk.make_tiny_config(), W)     # the tiny config only carries the tensor table here


@pytest.mark.parametrize("d,L,H,ffn,pre_ln,final_norm,T", CASES)
def test_oracle_transformer_matches_torch(orc, d, L, H, ffn, pre_ln, final_norm, T):
    W = make_weights("tf_.", d, L, ffn, final_norm, seed=d + T)
    x = np.random.default_rng(T).standard_normal((2, T, d)).astype(np.float32)
    got = oracle_model(orc, W).transformer_encoder(x, "tf_.", L, H, pre_ln, final_norm)
    want = torch_transformer(W, "tf_.", x, L, H, pre_ln, final_norm)
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("d,L,H,ffn,pre_ln,final_norm,T", CASES)
def test_gpu_transformer_bit_identical(orc, tmp_path, d, L, H, ffn, pre_ln, final_norm, T):
    from parakeet_cpp_amd import capi
    import gpu_common as G
    W = make_weights("tf_.", d, L, ffn, final_norm, seed=d + T)
    wp = os.path.join(str(tmp_path), "tf.safetensors")
    synth.save_weights(wp, W)
    x = np.random.default_rng(T).standard_normal((2, T, d)).astype(np.float32)
    want = oracle_model(orc, W).transformer_encoder(x, "tf_.", L, H, pre_ln, final_norm)
    tf = capi.Transformer(wp, "tf_.", d, L, H, ffn, pre_ln, final_norm)
    G.assert_bits_equal(tf.forward(x), want, f"transformer d={d} H={H} pre_ln={pre_ln}")
    tf.close()
