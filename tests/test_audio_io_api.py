"""Host-side audio entry points added for the facade's audio_io.hpp (reference include/parakeet/audio_io.hpp:24-39): an encoded
WAV image in memory decodes exactly like the same file on disk (pk_read_audio_memory == pk_read_audio, the latter pinned against
the real reference object in test_audio_vs_reference.py), the header-only info walk reports what the decoder finds, and
non-WAV / truncated input fails loudly."""
import struct
import wave

import numpy as np
import pytest

from parakeet_cpp_amd import capi


def write_wav(path, pcm16, rate, channels):
    with wave.open(path, "wb") as wf:
        wf.setnchannels(channels); wf.setsampwidth(2); wf.setframerate(rate)
        wf.writeframes(struct.pack(f"<{pcm16.size}h", *pcm16.reshape(-1).tolist()))


@pytest.mark.parametrize("rate,channels,frames", [(16000, 1, 9000), (44100, 2, 12345), (8000, 3, 801)])
def test_memory_image_equals_file_and_info(tmp_path, rate, channels, frames):
    rng = np.random.default_rng(rate + channels)
    pcm16 = (rng.standard_normal((frames, channels)) * 6000).clip(-32768, 32767).astype(np.int16)
    path = str(tmp_path / "a.wav")
    write_wav(path, pcm16, rate, channels)
    want, orig = capi.read_audio(path, 16000)
    got, sr, ch = capi.read_audio_memory(open(path, "rb").read(), 16000)
    assert sr == orig == rate and ch == channels
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert capi.audio_info(path) == (rate, channels, frames)              # get_audio_duration = frames / rate, no decode


def test_bad_images_fail_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="RIFF/WAVE"):
        capi.read_audio_memory(b"fLaC" + b"\0" * 64)
    with pytest.raises(RuntimeError, match="RIFF/WAVE"):
        capi.read_audio_memory(b"RIFF")
    p = str(tmp_path / "x.wav")
    open(p, "wb").write(b"RIFF\x24\0\0\0WAVEjunk")
    with pytest.raises(RuntimeError, match="decode WAV"):
        capi.audio_info(p)
    with pytest.raises(RuntimeError, match="open audio file"):
        capi.audio_info(str(tmp_path / "missing.wav"))
