"""Host-side audio ingestion against the REAL reference: oracle/Makefile compiles /root/reference/src/audio_io.cpp where it lies
(dr_wav decode, downmix_to_mono, sinc_resample, read_audio) against a 40-line stand-in for the one axiom type it wraps its output
in, into oracle/_ref/libpk_ref_audio.so.  The product's pk_resample / pk_read_audio (csrc/wav.cpp) must reproduce it: the
resampler is fp64 arithmetic rounded to fp32 at the end, so the comparison is (near-)exact: <= 1 fp32 ulp at unit scale (6e-8).
Needs no GPU (pure host code of libparakeet_amd.so)."""
import ctypes as C
import os
import struct
import wave

import numpy as np
import pytest

from conftest import ROOT
from parakeet_cpp_amd import capi, synth

REF = os.path.join(ROOT, "oracle", "_ref", "libpk_ref_audio.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libpk_ref_audio.so not built (reference tree absent at build time)")
f32p = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(REF)
    L.ref_resample.restype = f32p
    L.ref_resample.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    L.ref_read_audio.restype = f32p
    L.ref_read_audio.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_audio_free.argtypes = [f32p]
    return L


def ref_resample(L, x, src, dst):
    x = np.ascontiguousarray(x, np.float32)
    n = C.c_longlong(0)
    p = L.ref_resample(x.ctypes.data_as(f32p), x.size, src, dst, C.byref(n))
    assert n.value >= 0
    out = np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[: n.value].copy()
    L.ref_audio_free(p)
    return out


@pytest.mark.parametrize("src,dst,n", [(44100, 16000, 30000), (48000, 16000, 48000), (8000, 16000, 8000), (22050, 16000, 5000),
                                       (16000, 16000, 1000), (16000, 8000, 4001), (11025, 16000, 17)])
def test_resampler_matches_reference(ref, src, dst, n):
    x = synth.synth_pcm(1, n, seed=src + n)[0]
    want = ref_resample(ref, x, src, dst)
    got = capi.resample(x, src, dst)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 6e-8


def write_wav(path, pcm16, rate, channels):
    with wave.open(path, "wb") as wf:
        wf.setnchannels(channels); wf.setsampwidth(2); wf.setframerate(rate)
        wf.writeframes(struct.pack(f"<{pcm16.size}h", *pcm16.reshape(-1).tolist()))


@pytest.mark.parametrize("rate,channels", [(16000, 1), (44100, 2), (48000, 3), (8000, 1)])
def test_read_audio_matches_reference(ref, tmp_path, rate, channels):
    """read_audio(path) -> decode, mono downmix (sum * 1/channels), resample to 16 kHz."""
    rng = np.random.default_rng(rate + channels)
    pcm16 = (rng.standard_normal((9000, channels)) * 6000).clip(-32768, 32767).astype(np.int16)
    path = str(tmp_path / "a.wav")
    write_wav(path, pcm16, rate, channels)
    n, orig, ch = C.c_longlong(0), C.c_int(0), C.c_int(0)
    p = ref.ref_read_audio(path.encode(), 16000, C.byref(n), C.byref(orig), C.byref(ch))
    assert n.value > 0 and orig.value == rate and ch.value == channels
    want = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    ref.ref_audio_free(p)
    got, sr = capi.read_audio(path, 16000)
    assert sr == rate and got.shape == want.shape
    assert np.abs(got - want).max() <= 6e-8
