#!/usr/bin/env python3
"""Generate tests/golden/ref_compare_features_seed7.npz by RUNNING the reference author's mel-feature check,
/root/reference/scripts/compare_features.py (his PyTorch restatement of src/audio.cpp:100-158), on a seeded synthetic WAV.

The script is executed in place (never copied) inside a scratch directory holding the files it opens:
    models/2086-149220-0033.wav     <- 7.43 s of seeded PCM16 (744 frames: the script reshapes its C++ dump to (1,744,80))
    models/debug_features_cpp.bin   <- zeros of that shape (the script's C++-vs-Python diff is then meaningless and ignored)
torchaudio is not installed here; the script uses exactly one function of it, torchaudio.functional.melscale_fbanks
(norm="slaney", mel_scale="slaney").  A stub module provides that function, restating torchaudio's published algorithm
(torchaudio/functional/functional.py: linspace of FFT bin frequencies, mel points, triangle = min(down-slope, up-slope)
clamped at 0, Slaney area normalisation 2/(f[m+2]-f[m])) in float32 like the original.  Everything else -- pre-emphasis,
the WINDOW PLACEMENT (the script left-aligns the 400-tap Hann window in the 512-point frame: switch A1 of SURVEY.md 8c),
torch.stft(center, reflect), |X|^2, log(+2^-24), per-bin mean / unbiased std, /(std+1e-5) -- is the reference author's code.
usage (in the build container): python tools/make_golden_mel_from_reference.py"""
import contextlib
import io
import math
import os
import struct
import sys
import tempfile
import types
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
REF_SCRIPT = "/root/reference/scripts/compare_features.py"
OUT = os.path.join(ROOT, "tests", "golden", "ref_compare_features_seed7.npz")
N_SAMPLES = 743 * 160 + 37          # 1 + n/160 = 744 frames


def torchaudio_stub():
    import torch

    def hz_to_mel(f):
        f_sp = 200.0 / 3
        if f >= 1000.0:
            return (1000.0 - 0.0) / f_sp + math.log(f / 1000.0) / (math.log(6.4) / 27.0)
        return f / f_sp

    def mel_to_hz(m):
        f_sp = 200.0 / 3
        freqs = f_sp * m
        min_log_mel = 1000.0 / f_sp
        logstep = math.log(6.4) / 27.0
        log_t = m >= min_log_mel
        freqs[log_t] = 1000.0 * torch.exp(logstep * (m[log_t] - min_log_mel))
        return freqs

    def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
        assert norm == "slaney" and mel_scale == "slaney"
        all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
        m_pts = torch.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2)
        f_pts = mel_to_hz(m_pts)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
        up = slopes[:, 2:] / f_diff[1:]
        fb = torch.max(torch.zeros(1), torch.min(down, up))
        fb *= (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
        return fb

    ta = types.ModuleType("torchaudio")
    ta.functional = types.ModuleType("torchaudio.functional")
    ta.functional.melscale_fbanks = melscale_fbanks
    return ta


def main():
    import pkload
    pkload.load()
    from parakeet_cpp_amd import synth
    pcm = synth.synth_pcm(1, N_SAMPLES, seed=7)[0]
    pcm16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype(np.int16)
    sys.modules["torchaudio"] = torchaudio_stub()
    sys.modules["torchaudio.functional"] = sys.modules["torchaudio"].functional
    src = open(REF_SCRIPT).read()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "models"))
        with wave.open(os.path.join(td, "models", "2086-149220-0033.wav"), "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
            wf.writeframes(struct.pack(f"<{len(pcm16)}h", *pcm16.tolist()))
        np.zeros((1, 744, 80), np.float32).tofile(os.path.join(td, "models", "debug_features_cpp.bin"))
        ns = {"__name__": "__ref_compare_features__"}
        os.chdir(td)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                exec(compile(src, REF_SCRIPT, "exec"), ns)
        finally:
            os.chdir(cwd)
    feats = ns["features_py"].numpy().astype(np.float32)
    logmel = ns["log_mel"].numpy().astype(np.float32)
    assert feats.shape == (1, 744, 80), feats.shape
    np.savez_compressed(OUT, pcm16=pcm16, features=feats, log_mel=logmel)
    print(f"wrote {OUT}: features {feats.shape} range [{feats.min():.3f}, {feats.max():.3f}]")


if __name__ == "__main__":
    main()
