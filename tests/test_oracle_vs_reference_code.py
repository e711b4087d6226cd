"""The CPU oracle (oracle/pk_oracle.c) against the REFERENCE'S OWN CODE.

oracle/_ref/libpk_ref_model.so is /root/reference/src/{audio,encoder,lstm,rnnt,tdt,ctc,tdt_ctc,transformer,streaming_encoder,eou,
nemotron,sortformer,phrase_boost,vocab,timestamp}.cpp + include/parakeet/transcribe.hpp compiled where they lie (oracle/Makefile)
against a CPU stand-in for the un-vendored `axiom` tensor library (oracle/axiom_stub/).  Every decode loop, state revert,
duration skip, cache rotation, weight-name registration and tensor reshape below is therefore executed by the reference's
object code; only the tensor primitives underneath (matmul, conv, softmax, stft ...) are the stand-in's.

Bars: token ids / frames / lengths identical; floating-point stages within the stated tolerance (the oracle evaluates exp / log /
tanh with its own polynomials and sums in the canonical `sum64` order, the stand-in uses libm and double sums).
"""
import dataclasses

import numpy as np
import pytest

from conftest import pk
from parakeet_cpp_amd import synth

refmodel = pytest.importorskip("refmodel")
pytestmark = pytest.mark.skipif(not refmodel.available(), reason="oracle/_ref/libpk_ref_model.so not built (needs /root/reference)")


def enc_like(B, T, d, seed):
    x = np.random.default_rng(seed).standard_normal((B, T, d)).astype(np.float32)
    return (x - x.mean(-1, keepdims=True)) / x.std(-1, keepdims=True)


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


class Pair:
    """The same synthetic safetensors file loaded by the oracle and by the reference's model class."""

    def __init__(self, tmp, cfg, seed=42, kind=None, tweak=None, **kw):
        self.cfg = cfg
        self.W = synth.synth_weights(cfg, seed=seed)
        if tweak:
            tweak(self.W)
        self.wp = str(tmp / f"{cfg.name}_{seed}.safetensors")
        synth.save_weights(self.wp, self.W)
        import oracle
        self.om = oracle.Model(cfg, self.W)
        self.rm = refmodel.Model(cfg, self.wp, kind=kind, **kw)


@pytest.fixture(scope="module")
def tmp(tmp_path_factory):
    return tmp_path_factory.mktemp("refcode")


@pytest.fixture(scope="module")
def tiny(tmp, orc):
    return Pair(tmp, pk.make_tiny_config())


@pytest.fixture(scope="module")
def heads_110m(tmp, orc):
    """tdt-ctc-110m with ONE encoder layer: the full-size prediction net, joint, CTC head and subsampling."""
    return Pair(tmp, dataclasses.replace(pk.make_110m_config(), num_layers=1, name="110m-1L"))


def same_decode(o, r, conf_tol=2e-6, end=True):
    B = len(r.ids)
    for b in range(B):
        n = int(o["lens"][b])
        assert n == len(r.ids[b]), (b, n, len(r.ids[b]))
        assert np.array_equal(o["ids"][b, :n], r.ids[b]), b
        if r.start is not None:
            assert np.array_equal(o["start"][b, :n], r.start[b]), b
            if end:
                assert np.array_equal(o["end"][b, :n], r.end[b]), b
            assert np.max(np.abs(o["conf"][b, :n] - r.conf[b]), initial=0.0) <= conf_tol, b


# ───────────────────────── weight names (a15) ─────────────────────────
@pytest.mark.parametrize("preset,kind", [("tdt-ctc-110m", None), ("tdt-600m", None), ("rnnt-600m", None), ("nemotron-600m", "nemotron"),
                                         ("eou-120m", "eou")])
def test_reference_modules_find_every_synthetic_weight(tmp, preset, kind):
    """The names the reference registers through AX_REGISTER_* are the on-disk contract.  Loading our synthetic
    file into the reference's own module tree must leave NO parameter unset; the only tensor the reference ignores is
    `<joint>.pred_proj_.bias` (its Linear(bias=false), tdt.cpp:10-11 / rnnt.cpp:33 -- switch A5)."""
    cfg = dataclasses.replace(pk.PRESETS[preset](), num_layers=2, name=preset + "-2L")
    W = synth.synth_weights(cfg, seed=1)
    wp = str(tmp / f"names_{preset}.safetensors")
    synth.save_weights(wp, W)
    missing, unexpected = refmodel.Model(cfg, wp, kind=kind).load_report()
    assert missing == []
    assert unexpected == [cfg.joint_prefix + "pred_proj_.bias"]


def test_reference_sortformer_finds_every_synthetic_weight(tmp):
    from test_sortformer import tiny_sf
    sf = tiny_sf()
    wp = str(tmp / "names_sf.safetensors")
    synth.save_weights(wp, synth.synth_sortformer_weights(sf, seed=1))
    missing, unexpected = refmodel.Model(sf.nest_encoder, wp, kind="sortformer", sortformer=sf).load_report()
    assert missing == [] and unexpected == []


# ───────────────────────── front end (a1, a2, a4) ─────────────────────────
@pytest.mark.parametrize("n_mels,n", [(80, 16000), (128, 24000), (80, 5433), (80, 160000)])
@pytest.mark.parametrize("centered", [False, True])
def test_mel_matches_reference_preprocess_audio(orc, n_mels, n, centered):
    """preprocess_audio (src/audio.cpp:100-158) run by the reference: pre-emphasis, framing, power, the fp64 Slaney filterbank
    (audio.cpp:40-94 -- the reference's own code), log guard, unbiased per-bin normalisation, transpose."""
    pcm = synth.synth_pcm(1, n, seed=5)[0]
    refmodel.set_window_centered(centered)
    try:
        want = refmodel.preprocess_audio(pcm, n_mels=n_mels)
    finally:
        refmodel.set_window_centered(False)
    got = orc.mel(pcm, n_mels=n_mels, window_centered=centered)
    assert got.shape == want.shape == (1 + n // 160, n_mels)
    assert np.max(np.abs(got - want)) < 1e-3        # log amplifies round-off in near-empty bins; typical 3e-5


def test_pos_emb_is_the_reference_table(orc):
    for T, d in ((126, 512), (7, 64), (376, 1024)):
        want = refmodel.pos_emb(T, d)
        got = orc.pos_emb(T, d)
        assert np.max(np.abs(got - want)) < 2e-6    # std::sin / std::exp of the build's libm vs the oracle's float evaluation


# ───────────────────────── encoder (a3, a5-a8) ─────────────────────────
def test_subsampling_matches_reference_tiny(tiny):
    feats = np.random.default_rng(0).standard_normal((2, 203, tiny.cfg.mel_bins)).astype(np.float32)
    got, want = tiny.om.subsampling(feats), tiny.rm.subsampling(feats)
    assert got.shape == want.shape == (2, 26, tiny.cfg.hidden_size)
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("Tm", [1001, 64, 9])
def test_subsampling_matches_reference_full_width(heads_110m, Tm):
    """ConvSubsampling::forward (src/encoder.cpp:219-241) at the real geometry: 256 channels, 80 mel bins, conv / depthwise /
    pointwise stack with ReLU, the permute(0,2,1,3) flatten order and the 2560 -> 512 projection."""
    feats = np.random.default_rng(Tm).standard_normal((1, Tm, 80)).astype(np.float32)
    got, want = heads_110m.om.subsampling(feats), heads_110m.rm.subsampling(feats)
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-5


def test_subsampling_128_mel_bins(tmp, orc):
    p = Pair(tmp, dataclasses.replace(pk.make_tdt_600m_config(), num_layers=1, name="600m-1L"))
    feats = np.random.default_rng(3).standard_normal((1, 301, 128)).astype(np.float32)
    assert rel_err(p.om.subsampling(feats), p.rm.subsampling(feats)) < 1e-5


@pytest.mark.parametrize("layer", [0, 1])
def test_conformer_block_matches_reference(tiny, layer):
    x = np.random.default_rng(1).standard_normal((2, 37, tiny.cfg.hidden_size)).astype(np.float32)
    assert rel_err(tiny.om.conformer_block(layer, x), tiny.rm.conformer_block(layer, x)) < 2e-5


def test_full_width_block_matches_reference(heads_110m):
    x = np.random.default_rng(2).standard_normal((1, 126, 512)).astype(np.float32)
    assert rel_err(heads_110m.om.conformer_block(0, x), heads_110m.rm.conformer_block(0, x)) < 2e-5


def test_encoder_matches_reference(tiny, orc):
    pcm = synth.synth_pcm(3, 32000, seed=9)
    feats = np.stack([orc.mel(p) for p in pcm])
    assert rel_err(tiny.om.encoder(feats), tiny.rm.encoder(feats)) < 5e-5


@pytest.mark.slow
def test_encoder_110m_17_layers_matches_reference(tmp, orc):
    p = Pair(tmp, pk.make_110m_config())
    pcm = synth.synth_pcm(1, 160000, seed=1234)
    feats = np.stack([orc.mel(x) for x in pcm])
    got, want = p.om.encoder(feats), p.rm.encoder(feats)
    assert got.shape == want.shape == (1, 126, 512)
    assert rel_err(got, want) < 5e-5          # observed 2e-6


# ───────────────────────── CTC (a9, a10) ─────────────────────────
def test_ctc_head_and_greedy_match_reference(heads_110m, orc):
    enc = enc_like(3, 126, 512, 1)
    lp_o, lp_r = heads_110m.om.ctc_logprobs(enc), heads_110m.rm.ctc_logprobs(enc)
    assert np.max(np.abs(lp_o - lp_r)) < 2e-5
    o = orc.ctc_greedy(lp_o, 1024)
    same_decode(o, refmodel.ctc_greedy(lp_o, 1024, timestamps=True))
    same_decode(o, refmodel.ctc_greedy(lp_o, 1024, timestamps=False))
    assert o["lens"].sum() > 0, "degenerate test: nothing decoded"


# ───────────────────────── prediction net, joint (a11, a12) ─────────────────────────
@pytest.mark.parametrize("fx", ["tiny", "heads_110m"])
def test_lstm_step_and_joint_match_reference(request, fx, orc):
    """LSTMCell gate order i,f,g,o / merged bias / hidden_proj without bias (src/lstm.cpp:11-29), Embedding lookup, and the
    joint (src/tdt.cpp:15-24): compared through the first decode step's label log-probs, which the oracle can tap."""
    p = request.getfixturevalue(fx)
    cfg = p.cfg
    enc = enc_like(2, 5, cfg.hidden_size, 11)
    o = p.om.tdt_greedy(enc, first_logp=True)
    L, Hp = cfg.num_lstm_layers, cfg.pred_hidden
    for b in range(2):
        pred, _, _ = p.rm.prediction_step(cfg.blank_id, np.zeros((L, Hp), np.float32), np.zeros((L, Hp), np.float32))
        lab, dur = p.rm.joint(enc[b, 0], pred)
        assert np.max(np.abs(lab - o["first_logp"][b])) < 2e-5
        assert abs(float(np.logaddexp.reduce(dur.astype(np.float64)))) < 1e-5      # a log-softmax


# ───────────────────────── TDT / RNNT greedy loops (a13, a14) ─────────────────────────
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tdt_greedy_matches_reference_tiny(tiny, seed):
    enc = enc_like(4, 57, tiny.cfg.hidden_size, seed)
    o = tiny.om.tdt_greedy(enc)
    assert not o["overflow"] and o["lens"].sum() > 0
    same_decode(o, tiny.rm.tdt_greedy(enc, timestamps=True))
    same_decode(o, tiny.rm.tdt_greedy(enc, timestamps=False))      # the reference's e2e invariant: ids with == ids without


def test_tdt_greedy_matches_reference_110m_heads(heads_110m):
    enc = enc_like(3, 126, 512, 7)
    o = heads_110m.om.tdt_greedy(enc)
    assert o["lens"].min() > 5
    same_decode(o, heads_110m.rm.tdt_greedy(enc, timestamps=True))


def test_tdt_duration_zero_runs_match_reference(tmp, orc):
    """Inner `for sym < max_symbols` loop (src/tdt.cpp:66-105): a duration head biased towards 0 makes several symbols come out
    on one frame; ids, frames and the end-frame clamp have to follow the reference exactly."""
    def tweak(W):
        W["tdt_joint_.duration_proj_.bias"][0] += 1.2
        W["tdt_joint_.label_proj_.bias"][64] -= 1.0
    p = Pair(tmp, pk.make_tiny_config(name="tiny-dur0"), seed=8, tweak=tweak)
    enc = enc_like(4, 40, p.cfg.hidden_size, 5)
    o = p.om.tdt_greedy(enc, max_steps=4000)
    assert not o["overflow"]
    r = p.rm.tdt_greedy(enc, timestamps=True)
    same_decode(o, r)
    multi = sum(int(np.sum(np.diff(s) == 0)) for s in r.start)
    assert multi >= 3, "degenerate test: no frame emitted more than one symbol"


def test_tdt_two_lstm_layers_matches_reference(tmp, orc):
    cfg = dataclasses.replace(pk.make_tdt_600m_config(), num_layers=1, name="600m-heads")
    p = Pair(tmp, cfg, seed=3)
    enc = enc_like(2, 60, 1024, 4)
    o = p.om.tdt_greedy(enc)
    assert o["lens"].sum() > 0
    same_decode(o, p.rm.tdt_greedy(enc, timestamps=True, blank_id=cfg.blank_id))


def test_rnnt_greedy_matches_reference(tmp, orc):
    cfg = pk.make_tiny_config(name="tiny-rnnt", head="rnnt", durations=[], joint_prefix="joint_.", ctc_vocab_size=0, num_lstm_layers=2)
    p = Pair(tmp, cfg, seed=6)
    enc = enc_like(3, 45, cfg.hidden_size, 2)
    o = p.om.rnnt_greedy(enc)
    assert o["lens"].sum() > 0
    o["end"] = o["start"]                                     # rnnt.cpp:168: start == end == t
    same_decode(o, p.rm.rnnt_greedy(enc, timestamps=True))
    same_decode(o, p.rm.rnnt_greedy(enc, timestamps=False))


# ───────────────────────── phrase boosting (f4) ─────────────────────────
def test_trie_matches_reference(orc):
    rng = np.random.default_rng(0)
    phrases = [rng.integers(0, 40, rng.integers(1, 6)).tolist() for _ in range(25)]
    to, tr = orc.Trie(phrases), refmodel.Trie(phrases)
    assert to.size() == tr.size()
    states = {0}
    for tok in rng.integers(0, 40, 200).tolist():
        assert to.boosted_tokens(states, 64) == tr.boosted_tokens(states, 64)
        so, sr = to.advance(states, tok), tr.advance(states, tok)
        assert so == sr
        states = so


@pytest.mark.parametrize("boost", [0.0, 1.5, 5.0])
def test_boosted_ctc_matches_reference(orc, boost):
    rng = np.random.default_rng(3)
    lp = np.log(rng.dirichlet(np.ones(33) * 0.3, (2, 50))).astype(np.float32)
    phrases = [[3, 4, 5], [3, 9], [12], [7, 7, 8]]
    o = orc.ctc_greedy_boosted(lp, 32, orc.Trie(phrases), boost)
    same_decode(o, refmodel.ctc_greedy_boosted(lp, 32, refmodel.Trie(phrases), boost, timestamps=True))
    same_decode(o, refmodel.ctc_greedy_boosted(lp, 32, refmodel.Trie(phrases), boost, timestamps=False))


@pytest.mark.parametrize("boost", [0.0, 2.0, 6.0])
def test_boosted_tdt_matches_reference(tiny, orc, boost):
    """tdt_greedy_decode(_with_timestamps)_boosted (src/phrase_boost.cpp:177-350): boosted argmax, raw-log-prob confidence, trie
    advanced on every emission."""
    enc = enc_like(3, 50, tiny.cfg.hidden_size, 9)
    base = tiny.om.tdt_greedy(enc)
    seen = [int(t) for b in range(3) for t in base["ids"][b, :base["lens"][b]]]
    phrases = [seen[0:2], seen[3:6], [seen[1], (seen[2] + 1) % 64], [5, 6, 7]]
    o = tiny.om.tdt_greedy_boosted(enc, orc.Trie(phrases), boost)
    same_decode(o, tiny.rm.tdt_greedy_boosted(enc, refmodel.Trie(phrases), boost, timestamps=True))
    same_decode(o, tiny.rm.tdt_greedy_boosted(enc, refmodel.Trie(phrases), boost, timestamps=False))
    if boost == 6.0:
        assert any(not np.array_equal(o["ids"][b, :o["lens"][b]], base["ids"][b, :base["lens"][b]]) for b in range(3)), \
            "degenerate test: the boost changed nothing"


# ───────────────────────── Transformer (a16) and Sortformer (f4) ─────────────────────────
@pytest.mark.parametrize("d,L,H,ffn,pre_ln,final_norm,T", [(96, 2, 4, 192, False, False, 33), (192, 3, 8, 768, True, True, 50)])
def test_transformer_matches_reference(tmp, orc, d, L, H, ffn, pre_ln, final_norm, T):
    from test_transformer import make_weights
    W = make_weights("tf_.", d, L, ffn, final_norm, seed=d + T)
    wp = str(tmp / f"tf_{d}_{L}.safetensors")
    synth.save_weights(wp, W)
    x = np.random.default_rng(1).standard_normal((2, T, d)).astype(np.float32)
    want = refmodel.transformer_forward(wp, "tf_.", x, d, L, H, ffn, pre_ln, final_norm)
    got = orc.Model(pk.make_tiny_config(), W).transformer_encoder(x, "tf_.", L, H, pre_ln, final_norm)
    assert rel_err(got, want) < 2e-5


def test_sortformer_forward_and_segments_match_reference(tmp, orc):
    from test_sortformer import feats_like, tiny_sf
    sf = tiny_sf()
    W = synth.synth_sortformer_weights(sf, seed=5)
    wp = str(tmp / "sf.safetensors")
    synth.save_weights(wp, W)
    rm = refmodel.Model(sf.nest_encoder, wp, kind="sortformer", sortformer=sf)
    om = orc.Model(sf.nest_encoder, W)
    feats = feats_like(2, 97, 128, 1)
    got, want = om.sortformer_forward(feats, sf), rm.sortformer_forward(feats)
    assert got.shape == want.shape == (2, 13, 4)
    assert np.max(np.abs(got - want)) < 5e-5
    segs_r = rm.sortformer_diarize(feats[:1])
    segs_o = orc.probs_to_segments(got[0], sf.activity_threshold)
    margin = np.min(np.abs(want[0] - sf.activity_threshold))
    if margin > 1e-4:
        assert segs_o == segs_r
    assert len(segs_r) > 0


# ───────────────────────── streaming (f3) ─────────────────────────
@pytest.mark.parametrize("left,right,chunk", [(10, 1, 2560), (70, 0, 2560), (6, 0, 4000), (70, 13, 1999)])
def test_stream_matches_reference_chunk_by_chunk(tmp, orc, left, right, chunk):
    """StreamingAudioPreprocessor::process_chunk (src/audio.cpp:195-259), StreamingFastConformerEncoder::forward_chunk
    (src/streaming_encoder.cpp:430-472: leftover mel frames, K/V and conv caches, un-shifted position scores, context mask) and
    rnnt_streaming_decode_chunk (src/eou.cpp:17-98) with carried state, the reference's objects on one side, the oracle's
    `Stream` on the other, fed the same ragged PCM chunks."""
    cfg = pk.make_tiny_config(num_layers=2, name=f"tiny-stream-{left}-{right}", ctc_vocab_size=0, joint_prefix="joint_.")
    W = synth.synth_weights(cfg, seed=5)
    wp = str(tmp / f"stream_{left}_{right}.safetensors")
    synth.save_weights(wp, W)
    om = orc.Model(cfg, W)
    so = orc.Stream(om, left, right)
    rm = refmodel.Model(cfg, wp, kind="nemotron", att_left=left, att_right=right)
    sr = refmodel.Stream(rm)
    pcm = synth.synth_pcm(1, chunk * 14, seed=left + chunk)[0]
    n_enc = n_tok = 0
    for i in range(14):
        seg = pcm[i * chunk:(i + 1) * chunk]
        mo, mr = so.mel(seg), sr.mel(seg)
        assert mo.shape == mr.shape
        if mo.shape[0] == 0:
            continue
        assert np.max(np.abs(mo - mr)) < 2e-3                  # un-normalised log-mel
        eo, er = so.encode(mo), sr.encode(mo)
        assert eo.shape == er.shape, f"chunk {i}"
        if eo.shape[0] == 0:
            continue
        assert np.max(np.abs(eo - er)) < 2e-4, f"chunk {i}"
        n_enc += eo.shape[0]
        do = so.decode(eo)
        ids, st, en, cf = sr.decode(eo, blank_id=cfg.blank_id)
        assert np.array_equal(do["ids"], ids), f"chunk {i}"
        assert np.array_equal(do["start"], st) and np.array_equal(do["end"], en), f"chunk {i}"
        assert np.max(np.abs(do["conf"] - cf), initial=0.0) < 2e-6
        n_tok += len(ids)
    assert n_enc >= 10 and n_tok > 0


def test_nemotron_transcriber_matches_oracle_stream(tmp, orc):
    """parakeet::NemotronTranscriber::transcribe_chunk (src/nemotron.cpp:24-52) -- the class BASELINE configs[4] runs -- fed raw PCM
    chunks, against the oracle's Stream.push on the same chunks: cumulative token ids and absolute frames identical."""
    cfg = pk.make_tiny_config(num_layers=2, name="tiny-nemotron", vocab_size=1025, blank_id=1024, ctc_vocab_size=0, joint_prefix="joint_.",
                              num_lstm_layers=2)
    W = synth.synth_weights(cfg, seed=21)
    wp, vp = str(tmp / "nemo.safetensors"), str(tmp / "nemo_vocab.txt")
    synth.save_weights(wp, W)
    synth.save_vocab(vp, synth.synth_vocab(cfg.vocab_size - 1))
    nt = refmodel.NemotronTranscriber(cfg, wp, vp, att_left=70, att_right=1)
    so = orc.Stream(orc.Model(cfg, W), 70, 1)
    pcm = synth.synth_pcm(1, 2560 * 40, seed=4)[0]
    ids_o, st_o, en_o = [], [], []
    for i in range(40):
        seg = pcm[i * 2560:(i + 1) * 2560]
        r = so.push(seg)
        if r is not None:
            ids_o += r["ids"].tolist(); st_o += r["start"].tolist(); en_o += r["end"].tolist()
        ids, st, en, cf, text = nt.push(seg)
        assert ids.tolist() == ids_o and st.tolist() == st_o and en.tolist() == en_o, f"chunk {i}"
    assert len(ids_o) > 5


def test_sortformer_chunks_match_reference(tmp, orc):
    from test_sortformer import feats_like, tiny_sf
    sf = tiny_sf()
    W = synth.synth_sortformer_weights(sf, seed=5)
    wp = str(tmp / "sf_chunk.safetensors")
    synth.save_weights(wp, W)
    rm = refmodel.Model(sf.nest_encoder, wp, kind="sortformer", sortformer=sf, att_left=sf.att_context_left, att_right=sf.att_context_right)
    sr = refmodel.Stream(rm)
    so = orc.Stream(orc.Model(sf.nest_encoder, W), sf.att_context_left, sf.att_context_right)
    feats = feats_like(1, 200, 128, 2)[0]
    pos, n = 0, 0
    for size in (40, 27, 8, 5, 64, 56):
        f = feats[pos:pos + size]
        pos += size
        probs = so.sortformer_chunk(f, sf)
        segs_r, order = sr.sortformer_chunk(f)
        if probs.shape[0] == 0:
            assert segs_r == []
            continue
        n += probs.shape[0]
        if np.min(np.abs(probs - sf.activity_threshold)) > 1e-4:
            assert orc.probs_to_segments(probs, sf.activity_threshold) == segs_r
    assert n >= 20


# ───────────────────────── end to end: the reference's Transcriber class ─────────────────────────
def test_transcriber_end_to_end_matches_oracle_pipeline(tmp, orc):
    """parakeet::Transcriber(weights, vocab, config).transcribe(samples, opts) (include/parakeet/transcribe.hpp:53-185), the
    reference's top-level call: mel -> encoder -> CTC / TDT greedy (+ timestamps) -> detokenise, against the oracle's stages run
    on the same PCM.  Token ids and frames identical, confidences within 1e-5."""
    # the Transcriber calls the decoders with their DEFAULT blank_id = 1024 (transcribe.hpp:143-166, tdt.hpp:71-74): the small test
    # model therefore keeps the real vocabulary size
    cfg = pk.make_tiny_config(name="tiny-e2e", vocab_size=1025, ctc_vocab_size=1025, blank_id=1024)
    W = synth.synth_weights(cfg, seed=12)
    wp, vp = str(tmp / "e2e.safetensors"), str(tmp / "e2e_vocab.txt")
    synth.save_weights(wp, W)
    pieces = synth.synth_vocab(cfg.vocab_size - 1)
    synth.save_vocab(vp, pieces)
    om = orc.Model(cfg, W)
    tr = refmodel.Transcriber(cfg, wp, vp)
    total = 0
    for n, seed in ((32000, 1), (48000, 2), (20011, 3)):
        pcm = synth.synth_pcm(1, n, seed=seed)[0]
        enc = om.encoder(orc.mel(pcm)[None])
        o = om.tdt_greedy(enc)
        k = int(o["lens"][0])
        r = tr.transcribe(pcm, "tdt", timestamps=True)
        assert np.array_equal(r["token_ids"], o["ids"][0, :k])
        assert np.array_equal(r["start"], o["start"][0, :k]) and np.array_equal(r["end"], o["end"][0, :k])
        assert np.max(np.abs(r["conf"] - o["conf"][0, :k]), initial=0.0) < 1e-5
        assert np.array_equal(tr.transcribe(pcm, "tdt")["token_ids"], o["ids"][0, :k])
        c = orc.ctc_greedy(om.ctc_logprobs(enc), cfg.blank_id)
        kc = int(c["lens"][0])
        rc = tr.transcribe(pcm, "ctc", timestamps=True)
        assert np.array_equal(rc["token_ids"], c["ids"][0, :kc])
        assert np.array_equal(rc["start"], c["start"][0, :kc]) and np.array_equal(rc["end"], c["end"][0, :kc])
        total += k + kc
    assert total > 0, "degenerate test: nothing decoded"


@pytest.mark.slow
def test_transcriber_110m_10s_clip(tmp, orc):
    """BASELINE configs[0]/[1] model: tdt-ctc-110m, one 10 s clip through the reference's Transcriber on the CPU, TDT and CTC."""
    cfg = pk.make_110m_config()
    W = synth.synth_weights(cfg, seed=42)
    wp, vp = str(tmp / "110m.safetensors"), str(tmp / "110m_vocab.txt")
    synth.save_weights(wp, W)
    synth.save_vocab(vp, synth.synth_vocab(cfg.vocab_size - 1))
    om = orc.Model(cfg, W)
    tr = refmodel.Transcriber(cfg, wp, vp)
    pcm = synth.synth_pcm(1, 160000, seed=1234)[0]
    enc = om.encoder(orc.mel(pcm)[None])
    o = om.tdt_greedy(enc)
    r = tr.transcribe(pcm, "tdt", timestamps=True)
    k = int(o["lens"][0])
    assert k > 20 and np.array_equal(r["token_ids"], o["ids"][0, :k])
    assert np.array_equal(r["start"], o["start"][0, :k]) and np.array_equal(r["end"], o["end"][0, :k])
    c = orc.ctc_greedy(om.ctc_logprobs(enc), cfg.blank_id)
    assert np.array_equal(tr.transcribe(pcm, "ctc")["token_ids"], c["ids"][0, :c["lens"][0]])
