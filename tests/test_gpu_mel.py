"""GPU parity: mel front end (pk_mel, replacing preprocess_audio src/audio.cpp:100-158) vs the oracle,
bit-for-bit, at test sizes and at BASELINE's 10 s / 30 s clip lengths."""
import numpy as np
import pytest

import gpu_common as G
from conftest import pk
from parakeet_cpp_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_gpu(tmp_path_factory):
    from parakeet_cpp_amd import capi
    cfg = pk.make_tiny_config()
    p = tmp_path_factory.mktemp("w") / "tiny.safetensors"
    synth.save_weights(str(p), synth.synth_weights(cfg, seed=42))
    return capi.Model(str(p), cfg, device=0)


@pytest.mark.parametrize("n", [16000, 5433, 160000, 257])
def test_mel_bit_identical(tiny_gpu, orc, n):
    pcm = synth.synth_pcm(3, n, seed=11)
    feats, lm = tiny_gpu.mel(pcm, return_logmel=True)
    for b in range(3):
        of, ol = orc.mel(pcm[b], return_logmel=True)
        assert np.array_equal(lm[b].view(np.uint32), ol.view(np.uint32)), "log-mel differs"
        assert np.array_equal(feats[b].view(np.uint32), of.view(np.uint32)), "features differ"


def test_mel_zeros_shape_and_determinism(tiny_gpu):      # reference tests/test_all.cpp:727-753
    z = np.zeros((1, 16000), np.float32)
    a, b = tiny_gpu.mel(z), tiny_gpu.mel(z)
    assert a.shape == (1, 101, 80) and np.array_equal(a, b)


def test_mel_properties_full_batch(tiny_gpu):
    """BASELINE cfg-A size (64 x 10 s): per-bin mean 0 / unit unbiased variance (size-independent property)."""
    pcm = synth.synth_pcm(64, 160000, seed=1234)
    f = tiny_gpu.mel(pcm)
    assert f.shape == (64, 1001, 80)
    assert np.max(np.abs(f.mean(axis=1))) < 1e-3
    assert np.max(np.abs(f.std(axis=1, ddof=1) - 1.0)) < 1e-3


def test_mel_centered_window_switch_bit_identical(tmp_path_factory, orc):
    """Switch A1 (pk_config.stft_window_centered = 1): torch.stft / NeMo placement of the Hann window, also bit-for-bit."""
    import dataclasses
    from parakeet_cpp_amd import capi
    cfg = dataclasses.replace(pk.make_tiny_config(), stft_window_centered=True)
    p = tmp_path_factory.mktemp("wc") / "tiny.safetensors"
    synth.save_weights(str(p), synth.synth_weights(cfg, seed=42))
    gm = capi.Model(str(p), cfg, device=0)
    pcm = synth.synth_pcm(2, 16000, seed=12)
    feats = gm.mel(pcm)
    for b in range(2):
        want = orc.mel(pcm[b], window_centered=True)
        assert np.array_equal(feats[b].view(np.uint32), want.view(np.uint32))
        assert not np.array_equal(want, orc.mel(pcm[b]))
    gm.close()


@pytest.mark.parametrize("n_mels,normalize", [(80, True), (128, False)])
def test_standalone_frontend_matches_oracle(orc, n_mels, normalize):
    """pk_frontend_* = preprocess_audio without a model (the reference README's diarization usage feeds
    preprocess_audio(samples, {.normalize = false}) to Sortformer::diarize): bit-identical to the oracle's mel."""
    from parakeet_cpp_amd import capi
    fe = capi.Frontend(n_mels=n_mels, normalize=normalize)
    for n in (16000, 48123):
        pcm = synth.synth_pcm(1, n, seed=n)[0]
        G.assert_bits_equal(fe.features(pcm), orc.mel(pcm, n_mels=n_mels, normalize=normalize), f"frontend {n_mels} mels, normalize={normalize}")
    fe.close()
