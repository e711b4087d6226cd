"""GPU: the drop-in C++ surface.  examples/transcribe_wav.cpp is the reference README's usage verbatim
(parakeet::Transcriber t(weights, vocab); t.to_gpu(); t.transcribe("x.wav")) compiled against the header-only
facade; its token ids must equal the oracle's on the same 16 kHz WAV (BASELINE configs[0]: single WAV, CTC greedy,
here checked for both decoders), and its text must equal the detokenisation of those ids."""
import json
import os
import subprocess

import numpy as np
import pytest

import gpu_common as G
from conftest import ROOT, pk
from parakeet_cpp_amd import capi, synth

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "parakeet.cpp_amd", "examples", "transcribe_wav")


def test_transcriber_on_a_wav_matches_oracle(tmp_path, orc):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE)])
    cfg = pk.make_110m_config()
    W = synth.synth_weights(cfg, seed=42)
    wp, vp, ap = str(tmp_path / "model.safetensors"), str(tmp_path / "vocab.txt"), str(tmp_path / "clip.wav")
    synth.save_weights(wp, W)
    pieces = synth.synth_vocab(1024)
    synth.save_vocab(vp, pieces)
    pcm = synth.synth_pcm(1, 48000, seed=21)[0]
    synth.write_wav_pcm16(ap, pcm)
    q = (np.clip(pcm, -1, 1) * 32767.0).astype("<i2").astype(np.float32) / 32768.0    # what the WAV holds
    om = orc.Model(cfg, W)
    enc = om.encoder(np.stack([orc.mel(q)]))
    want = {"tdt": om.tdt_greedy(enc), "ctc": orc.ctc_greedy(om.ctc_logprobs(enc), 1024)}
    for dec in ("tdt", "ctc"):
        out = subprocess.run([EXE, wp, vp, ap, dec, "--timestamps"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        r = json.loads(out.stdout)
        n = want[dec]["lens"][0]
        ids = want[dec]["ids"][0, :n].tolist()
        assert r["token_ids"] == ids, dec
        text = "".join(pieces[i] for i in ids).replace("▁", " ")
        assert r["text"] == (text[1:] if text.startswith(" ") else text)
        assert n > 3 and len(r["words"]) >= 1
        starts = [w[1] for w in r["words"]]
        assert starts == sorted(starts)                      # monotonic word timestamps (tests/test_all.cpp:946-963)
    # Transcriber::to_all_gpus() (new, additive): replicas over RCCL, same answer
    out = subprocess.run([EXE, wp, vp, ap, "tdt", "--all-gpus"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    last = out.stdout.strip().splitlines()[-1]               # RCCL prints a version banner on stdout when its first communicator is made
    assert json.loads(last)["token_ids"] == want["tdt"]["ids"][0, :want["tdt"]["lens"][0]].tolist()
    # TranscribeOptions.boost_phrases through the facade (transcribe.hpp:41-42; CLI --boost / --boost-score, src/main.cpp:23-25)
    toks = want["tdt"]["ids"][0, :want["tdt"]["lens"][0]].tolist()
    def text_of(ids):
        t = "".join(pieces[i] for i in ids).replace("▁", " ")
        return t[1:] if t.startswith(" ") else t
    phrases = [text_of(toks[:2] + [77]), text_of([300, 301, 302])]
    gm = capi.Model(wp, cfg, vocab_path=vp)                                   # host-only handle: the tokenizer
    trie = orc.Trie([gm.tokenize(p) for p in phrases])
    gm.close()
    wb = om.tdt_greedy_boosted(enc, trie, 4.0)
    out = subprocess.run([EXE, wp, vp, ap, "tdt", "--boost", phrases[0], "--boost", phrases[1], "--boost-score", "4"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout)["token_ids"]
    assert got == wb["ids"][0, :wb["lens"][0]].tolist()
    assert got != toks, "degenerate test: the boost changed nothing"
    bad = subprocess.run([EXE, wp, str(tmp_path / "missing_vocab.txt"), ap], capture_output=True, text=True)
    assert bad.returncode == 1 and "Cannot open vocab file" in bad.stderr


def test_nemotron_transcriber_streams_a_wav(tmp_path, orc):
    """parakeet::NemotronTranscriber (reference include/parakeet/nemotron.hpp:54-133): transcribe_chunk / get_text /
    get_timestamped_tokens on 160 ms chunks of a WAV == the oracle's Stream.push on the same chunks (2-layer cut of the
    nemotron-600m architecture; blank id 1024 as the reference's transcribe_chunk decodes)."""
    import dataclasses
    exe = os.path.join(ROOT, "parakeet.cpp_amd", "examples", "stream_wav")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    cfg = dataclasses.replace(pk.make_nemotron_600m_config(), num_layers=2, name="nemotron-2L")
    W = synth.synth_weights(cfg, seed=9)
    wp, vp, ap = str(tmp_path / "model.safetensors"), str(tmp_path / "vocab.txt"), str(tmp_path / "clip.wav")
    synth.save_weights(wp, W)
    pieces = synth.synth_vocab(cfg.vocab_size - 1)
    synth.save_vocab(vp, pieces)
    pcm = synth.synth_pcm(1, 2560 * 40, seed=33)[0]
    synth.write_wav_pcm16(ap, pcm)
    q = (np.clip(pcm, -1, 1) * 32767.0).astype("<i2").astype(np.float32) / 32768.0
    st = orc.Stream(orc.Model(cfg, W), 70, 1)
    ids, frames = [], []
    for i in range(40):
        r = st.push(q[i * 2560:(i + 1) * 2560])
        if r is not None:
            ids += r["ids"].tolist(); frames += list(zip(r["start"].tolist(), r["end"].tolist()))
    out = subprocess.run([exe, wp, vp, ap, "2", "2560", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert [t[0] for t in r["tokens"]] == ids and [(t[1], t[2]) for t in r["tokens"]] == frames
    assert len(ids) > 0, "degenerate test: nothing decoded"
    text = "".join(pieces[i] for i in ids).replace("▁", " ")
    assert r["text"] == (text[1:] if text.startswith(" ") else text)


def test_diarized_transcriber_on_a_wav(tmp_path, orc):
    """parakeet::Sortformer / DiarizedTranscriber (reference include/parakeet/diarize.hpp:55-78, src/diarize.cpp:10-106) through
    examples/diarize_wav.cpp: segments == oracle Sortformer on the un-normalised 128-bin log-mel of the same WAV; every word gets the
    speaker with the largest total overlap (recomputed here from the printed segments and the ASR word times)."""
    exe = os.path.join(ROOT, "parakeet.cpp_amd", "examples", "diarize_wav")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    sf = pk.make_sortformer_117m_config()
    Wsf = synth.synth_sortformer_weights(sf, seed=11)
    cfg = pk.make_110m_config()
    W = synth.synth_weights(cfg, seed=42)
    sp, wp, vp, ap = (str(tmp_path / n) for n in ("sf.safetensors", "model.safetensors", "vocab.txt", "clip.wav"))
    synth.save_weights(sp, Wsf)
    synth.save_weights(wp, W)
    synth.save_vocab(vp, synth.synth_vocab(1024))
    pcm = synth.synth_pcm(1, 64000, seed=4)[0]
    synth.write_wav_pcm16(ap, pcm)
    q = (np.clip(pcm, -1, 1) * 32767.0).astype("<i2").astype(np.float32) / 32768.0
    probs = orc.Model(sf.nest_encoder, Wsf).sortformer_forward(orc.mel(q, n_mels=128, normalize=False)[None], sf)[0]
    want = orc.probs_to_segments(probs, 0.5)
    assert len(want) > 0
    out = subprocess.run([exe, sp, ap], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    segs = [(s[0], np.float32(s[1]), np.float32(s[2])) for s in json.loads(out.stdout)["segments"]]
    assert segs == [(s, np.float32(a), np.float32(b)) for s, a, b in want]
    out = subprocess.run([exe, sp, ap, wp, vp], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert [(s[0], np.float32(s[1]), np.float32(s[2])) for s in r["segments"]] == segs
    assert len(r["words"]) > 0
    for word, spk, a, b in r["words"]:
        ov = {}
        for s, sa, sb in segs:
            o = min(np.float32(b), sb) - max(np.float32(a), sa)
            if o > 0:
                ov[s] = ov.get(s, np.float32(0)) + o
        best = max(ov.values()) if ov else None
        assert (spk == -1 and not ov) or (spk in ov and ov[spk] == best), (word, spk, ov)


def test_parakeet_cli_modes(tmp_path, orc):
    """examples/parakeet_cli.cpp = the reference's CLI surface (src/main.cpp:12-37,642-727): tdt-ctc-110m with --ctc / --timestamps /
    --boost, sortformer, and the argument errors; token lines must equal the oracle's ids."""
    exe = os.path.join(ROOT, "parakeet.cpp_amd", "examples", "parakeet_cli")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    cfg = pk.make_110m_config()
    W = synth.synth_weights(cfg, seed=42)
    wp, vp, ap = str(tmp_path / "model.safetensors"), str(tmp_path / "vocab.txt"), str(tmp_path / "clip.wav")
    synth.save_weights(wp, W)
    synth.save_vocab(vp, synth.synth_vocab(1024))
    pcm = synth.synth_pcm(1, 48000, seed=21)[0]
    synth.write_wav_pcm16(ap, pcm)
    q = (np.clip(pcm, -1, 1) * 32767.0).astype("<i2").astype(np.float32) / 32768.0
    om = orc.Model(cfg, W)
    enc = om.encoder(np.stack([orc.mel(q)]))
    t = om.tdt_greedy(enc)
    c = orc.ctc_greedy(om.ctc_logprobs(enc), 1024)

    def tokens_of(stdout):
        line = [l for l in stdout.splitlines() if l.startswith("Tokens (")][0]
        return [int(x) for x in line.split("):")[1].split()]

    out = subprocess.run([exe, wp, ap, "--vocab", vp, "--timestamps", "--gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert tokens_of(out.stdout) == t["ids"][0, :t["lens"][0]].tolist()
    assert "--- Word timestamps ---" in out.stdout and "--- Transcription ---" in out.stdout
    out = subprocess.run([exe, wp, ap, "--vocab", vp, "--ctc"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert tokens_of(out.stdout) == c["ids"][0, :c["lens"][0]].tolist()
    bad = subprocess.run([exe, wp, ap, "--frobnicate"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Unknown option" in bad.stderr
    bad = subprocess.run([exe, wp, ap, "--model", "diarized", "--vocab", vp], capture_output=True, text=True)
    assert bad.returncode == 1 and "--sortformer-weights required" in bad.stderr
    sf = pk.make_sortformer_117m_config()
    sp = str(tmp_path / "sf.safetensors")
    synth.save_weights(sp, synth.synth_sortformer_weights(sf, seed=11))
    out = subprocess.run([exe, sp, ap, "--model", "sortformer"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "--- Speaker Segments (" in out.stdout
