"""Parity against the REFERENCE: tests/golden/ref_compare_encoder_110m_seed42.npz holds the outputs of the reference
author's own PyTorch restatement of the encoder (/root/reference/scripts/compare_encoder.py), executed by
tools/make_golden_from_reference.py on this repo's seeded synthetic tdt-ctc-110m weights.  Pinned here: the 17 Conformer
blocks (FeedForward, rel-pos attention + rel_shift, conv module, final norm), the CTC head, its argmax and the greedy
collapse.  Floating point: torch/MKL sums in a different order than our k-ordered fma chains, so activations are compared
with a stated tolerance (5e-4 absolute on values of magnitude ~4 after 17 layers; observed 2.3e-5; block 0 alone: 8e-6 against a 1e-4 bound); the CTC token ids
must be IDENTICAL (the fixture's smallest top-1/top-2 log-prob margin is 1.4e-2, far above that noise).
CPU test: the oracle (what every GPU parity test is checked against).  GPU test: the product, through the C ABI."""
import os

import numpy as np
import pytest

from conftest import pk
from parakeet_cpp_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_compare_encoder_110m_seed42.npz")
ATOL_LAYER0, ATOL_FINAL, ATOL_LOGP = 1e-4, 5e-4, 1e-3


@pytest.fixture(scope="module")
def gold():
    g = np.load(GOLD)
    assert int(g["weight_seed"]) == 42 and g["block_input"].shape == (1, 93, 512)
    return g


@pytest.fixture(scope="module")
def weights():
    return synth.synth_weights(pk.make_110m_config(), seed=42)


def check_against_gold(gold, layer0, final, logp, tokens):
    d0 = np.abs(layer0 - gold["layer0_out"]).max()
    d1 = np.abs(final - gold["encoder_out"]).max()
    assert d0 <= ATOL_LAYER0, f"ConformerBlock 0 differs from the reference restatement by {d0:.3e}"
    assert d1 <= ATOL_FINAL, f"17-layer encoder output differs from the reference restatement by {d1:.3e}"
    assert np.array_equal(logp.argmax(-1).astype(np.int32), gold["ctc_argmax"]), "per-frame CTC argmax differs from the reference"
    assert np.abs(logp.max(-1) - gold["ctc_best_logp"]).max() <= ATOL_LOGP
    assert list(tokens) == gold["ctc_tokens"].tolist(), "CTC greedy token ids differ from the reference"
    return d0, d1


def test_oracle_matches_reference_restatement(gold, weights, orc):
    om = orc.Model(pk.make_110m_config(), weights)
    x = gold["block_input"].copy()
    layer0 = om.conformer_block(0, x)
    final = layer0
    for l in range(1, 17):
        final = om.conformer_block(l, final)
    logp = om.ctc_logprobs(final)
    r = orc.ctc_greedy(logp, 1024)
    d0, d1 = check_against_gold(gold, layer0, final, logp, r["ids"][0, : r["lens"][0]])
    print(f"oracle vs reference restatement: layer0 max|d| {d0:.2e}, 17 layers {d1:.2e}")


@pytest.mark.gpu
def test_gpu_matches_reference_restatement(gold, weights, tmp_path):
    from parakeet_cpp_amd import capi
    cfg = pk.make_110m_config()
    wp = os.path.join(str(tmp_path), "w.safetensors")
    synth.save_weights(wp, weights)
    gm = capi.Model(wp, cfg, device=0)
    x = gold["block_input"].copy()
    layer0 = gm.conformer_blocks(x, 0, 1)
    final = gm.conformer_blocks(x)
    c = gm.ctc_decode(final, return_logp=True)
    check_against_gold(gold, layer0, final, c["logp"], c["ids"][0, : c["lens"][0]])
    gm.close()
