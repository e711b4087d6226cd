"""CPU-side checks of the drop-in boundary: libparakeet_amd.so loads without a GPU, exports every symbol
include/parakeet_amd.h declares, and every compute entry point FAILS LOUDLY (no CPU fallback) when no device
is present.  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pk
from parakeet_cpp_amd import capi, synth

HEADER = os.path.join(ROOT, "include", "parakeet_amd.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in include/parakeet_amd.h but not exported: {missing}"


def test_presets_match_reference_config_values():
    """include/parakeet/config.hpp:77-135 (the reference's ConfigPresets tests, tests/test_all.cpp:135-194)."""
    L = capi.lib()
    c = capi.PkConfig()
    capi.check(L.pk_config_preset(b"tdt-ctc-110m", C.byref(c)))
    assert (c.hidden_size, c.num_layers, c.num_heads, c.ffn_intermediate, c.mel_bins) == (512, 17, 8, 2048, 80)
    assert (c.vocab_size, c.num_lstm_layers, c.ctc_vocab_size, c.num_durations, c.blank_id) == (1025, 1, 1025, 5, 1024)
    assert list(c.durations)[:5] == [0, 1, 2, 3, 4] and c.joint_prefix == b"tdt_joint_."
    capi.check(L.pk_config_preset(b"tdt-600m", C.byref(c)))
    assert (c.hidden_size, c.num_layers, c.mel_bins, c.vocab_size, c.num_lstm_layers, c.ctc_vocab_size) == (1024, 24, 128, 8193, 2, 0)
    capi.check(L.pk_config_preset(b"rnnt-600m", C.byref(c)))
    assert (c.hidden_size, c.vocab_size, c.rnnt_head, c.num_durations) == (1024, 1025, 1, 0)
    assert L.pk_config_preset(b"nope", C.byref(c)) != 0
    for name, f in pk.PRESETS.items():
        if name == "tiny":
            continue
        capi.check(L.pk_config_preset(name.encode(), C.byref(c)))
        py = capi.to_pk_config(f())
        for fld, _ in capi.PkConfig._fields_:
            a, b = getattr(c, fld), getattr(py, fld)
            assert (list(a) == list(b)) if fld == "durations" else (a == b), (name, fld)


def test_frame_count_helpers():
    L = capi.lib()
    assert L.pk_mel_num_frames(160000) == 1001 and L.pk_mel_num_frames(480000) == 3001 and L.pk_mel_num_frames(16000) == 101
    assert L.pk_encoder_num_frames(1001) == 126 and L.pk_encoder_num_frames(3001) == 376


@pytest.mark.skipif(capi.device_count() > 0, reason="checks the no-GPU failure mode")
def test_no_gpu_is_a_loud_error(tmp_path):
    cfg = pk.make_tiny_config()
    wp = tmp_path / "t.safetensors"
    synth.save_weights(str(wp), synth.synth_weights(cfg))
    m = capi.Model(str(wp), cfg)                       # host-side load works without a GPU
    with pytest.raises(capi.PkError) as e:
        m.to_gpu(0)
    assert e.value.code == -4 and "no CPU path" in str(e.value)
    with pytest.raises(capi.PkError) as e:
        m.mel(np.zeros(16000, np.float32))
    assert e.value.code == -4
    with pytest.raises(capi.PkError):
        capi.diag_math("exp", [0.0])
    with pytest.raises(capi.PkError) as e:             # the multi-GPU entry point as well
        capi.Group(str(wp), cfg)
    assert e.value.code == -4


def test_strict_weight_loading_errors(tmp_path):
    cfg = pk.make_tiny_config()
    with pytest.raises(capi.PkError) as e:
        capi.Model(str(tmp_path / "missing.safetensors"), cfg)
    assert e.value.code == -2 and "Cannot open weights file" in str(e.value)
    bad = tmp_path / "bad.safetensors"
    bad.write_bytes(b"\xff" * 64)
    with pytest.raises(capi.PkError):
        capi.Model(str(bad), cfg)
    wp = tmp_path / "t.safetensors"
    synth.save_weights(str(wp), synth.synth_weights(cfg))
    with pytest.raises(capi.PkError) as e:             # reference: std::runtime_error("Cannot open vocab file: ...") vocab.cpp:12-14
        capi.Model(str(wp), cfg, vocab_path=str(tmp_path / "no_vocab.txt"))
    assert "Cannot open vocab file" in str(e.value)
