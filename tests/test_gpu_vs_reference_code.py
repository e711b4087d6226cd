"""GPU product (HIP, through the C ABI) directly against the REFERENCE'S OWN CODE (oracle/_ref/libpk_ref_model.so: the reference
sources compiled where they lie on the CPU stand-in for axiom -- see tests/test_oracle_vs_reference_code.py).  The library is
prebuilt in the authoring container and travels with the snapshot; nothing here reads /root/reference.

parakeet::Transcriber(weights, vocab, config).transcribe(samples, options) on the CPU  vs  pk_transcribe_pcm on the MI355X:
token ids, frames and text identical, confidences within 1e-5, for TDT and CTC, with and without timestamps and phrase boosting."""
import dataclasses

import numpy as np
import pytest

from conftest import pk
from parakeet_cpp_amd import synth

pytestmark = pytest.mark.gpu
refmodel = pytest.importorskip("refmodel")


def _need():
    if not refmodel.available():
        pytest.skip("oracle/_ref/libpk_ref_model.so not shipped")


def _pair(tmp, cfg, seed):
    from parakeet_cpp_amd import capi
    wp, vp = str(tmp / f"{cfg.name}.safetensors"), str(tmp / f"{cfg.name}_vocab.txt")
    synth.save_weights(wp, synth.synth_weights(cfg, seed=seed))
    synth.save_vocab(vp, synth.synth_vocab(cfg.vocab_size - 1))
    return capi.Model(wp, cfg, vocab_path=vp, device=0), refmodel.Transcriber(cfg, wp, vp)


def _same(g, r, timestamps):
    assert g["token_ids"] == r["token_ids"].tolist()
    assert g["text"] == r["text"]
    if timestamps:
        assert g["start"] == r["start"].tolist() and g["end"] == r["end"].tolist()
        assert np.max(np.abs(np.asarray(g["conf"], np.float32) - r["conf"]), initial=0.0) < 1e-5
        assert len(g["words"]) == r["n_words"]


def test_small_model_every_mode(tmp_path):
    _need()
    cfg = pk.make_tiny_config(name="tiny-vsref", vocab_size=1025, ctc_vocab_size=1025, blank_id=1024)   # the Transcriber's fixed blank id
    gm, tr = _pair(tmp_path, cfg, 12)
    n_tok = 0
    clips = [synth.synth_pcm(1, n, seed=s)[0] for n, s in ((32000, 1), (48000, 2), (20011, 3), (8000, 4))]
    for dec in ("tdt", "ctc"):
        for ts in (False, True):
            got = gm.transcribe_pcm(clips, decoder=dec, timestamps=ts)
            for c, g in zip(clips, got):
                r = tr.transcribe(c, dec, timestamps=ts)
                _same(g, r, ts)
                n_tok += len(g["token_ids"])
    assert n_tok > 0, "degenerate test: nothing decoded"


def test_small_model_boosted(tmp_path):
    _need()
    cfg = pk.make_tiny_config(name="tiny-vsref-boost", vocab_size=1025, ctc_vocab_size=1025, blank_id=1024)
    gm, tr = _pair(tmp_path, cfg, 13)
    clip = synth.synth_pcm(1, 40000, seed=5)[0]
    base = tr.transcribe(clip, "tdt")
    pieces = synth.synth_vocab(cfg.vocab_size - 1)
    # phrases made of pieces the model emits (so the trie is walked) plus one it does not
    phrases = ["".join(pieces[t] for t in base["token_ids"][i:i + 2]).replace("▁", " ").strip() for i in (0, 3)] + ["kato mire"]
    phrases = [p for p in phrases if p]
    for dec in ("tdt", "ctc"):
        for ts in (False, True):
            g = gm.transcribe_pcm([clip], decoder=dec, timestamps=ts, boost_phrases=phrases, boost_score=4.0)[0]
            _same(g, tr.transcribe(clip, dec, timestamps=ts, boost_phrases=phrases, boost_score=4.0), ts)


def test_tdt_ctc_110m_10s_clips(tmp_path):
    """BASELINE configs[0]/[1] model at full size: two 10 s clips, TDT with timestamps and CTC, product on the GPU vs the reference's
    Transcriber on the CPU."""
    _need()
    cfg = pk.make_110m_config()
    gm, tr = _pair(tmp_path, cfg, 42)
    clips = list(synth.synth_pcm(2, 160000, seed=1234))
    got = gm.transcribe_pcm(clips, decoder="tdt", timestamps=True)
    gotc = gm.transcribe_pcm(clips, decoder="ctc")
    for c, g, gc in zip(clips, got, gotc):
        _same(g, tr.transcribe(c, "tdt", timestamps=True), True)
        _same(gc, tr.transcribe(c, "ctc"), False)
        assert len(g["token_ids"]) > 20


def test_tdt_600m_shapes_one_clip(tmp_path):
    """TDTTranscriber (transcribe.hpp:200-299): 128 mel bins, d = 1024, hd = 128, two LSTM layers, vocab 8193 -- reduced to 3 layers
    so the CPU side stays in seconds.  NOTE the reference's TDTTranscriber decodes with the DEFAULT blank id 1024 (not 8192)."""
    _need()
    cfg = dataclasses.replace(pk.make_tdt_600m_config(), num_layers=3, name="600m-3L", blank_id=1024)
    gm, tr = _pair(tmp_path, cfg, 7)
    clip = synth.synth_pcm(1, 80000, seed=9)[0]
    g = gm.transcribe_pcm([clip], decoder="tdt", timestamps=True)[0]
    _same(g, tr.transcribe(clip, "tdt", timestamps=True), True)
