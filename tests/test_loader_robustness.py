"""The weight / vocabulary loaders at the boundary must refuse malformed files with an error, never crash: truncated and bit-flipped
safetensors images (header length, JSON, offsets), wrong shapes and dtypes, missing tensors, an unreadable vocabulary
(reference: std::runtime_error on unreadable vocab, src/vocab.cpp:12-14; the reference loads weights non-strictly -- we are strict)."""
import json
import struct

import numpy as np
import pytest

from conftest import pk
from parakeet_cpp_amd import capi, synth


@pytest.fixture(scope="module")
def image(tmp_path_factory):
    cfg = pk.make_tiny_config()
    p = str(tmp_path_factory.mktemp("ld") / "tiny.safetensors")
    synth.save_weights(p, synth.synth_weights(cfg, seed=42))
    return cfg, np.fromfile(p, np.uint8)


def test_good_image_loads(image):
    cfg, img = image
    capi.Model(img, cfg).close()


def test_truncations_and_corruptions_are_refused(image):
    cfg, img = image
    hlen = struct.unpack("<Q", img[:8].tobytes())[0]
    rng = np.random.default_rng(0)
    cases = [img[:4], img[:8], img[: 8 + hlen // 2], img[: 8 + hlen], img[: len(img) // 2], img[:-1]]
    bad_len = img.copy(); bad_len[:8] = np.frombuffer(struct.pack("<Q", 1 << 40), np.uint8); cases.append(bad_len)
    for _ in range(40):                                   # random byte flips inside the JSON header
        c = img.copy()
        for pos in rng.integers(8, 8 + hlen, size=3):
            c[pos] = rng.integers(0, 256)
        cases.append(c)
    refused = 0
    for c in cases:
        try:
            capi.Model(c, cfg).close()                    # a flip may land in whitespace / a digit that keeps the file valid
        except RuntimeError:
            refused += 1
    assert refused >= 7 + 20                              # every structural case and most header flips


def test_wrong_shape_dtype_and_missing_tensor(image, tmp_path):
    cfg, img = image
    hlen = struct.unpack("<Q", img[:8].tobytes())[0]
    hdr = json.loads(img[8:8 + hlen].tobytes())
    body = img[8 + hlen:].tobytes()

    def rebuild(h):
        hb = json.dumps(h).encode()
        return np.frombuffer(struct.pack("<Q", len(hb)) + hb + body, np.uint8)

    name = "encoder_.layers_.0.ffn1_.fc1_.weight"
    h = json.loads(json.dumps(hdr)); h[name]["shape"] = h[name]["shape"][::-1]
    m = capi.Model(rebuild(h), cfg)                       # parses (same element count); the shape check happens at upload time on a GPU
    m.close()
    h = json.loads(json.dumps(hdr)); h[name]["dtype"] = "F16"
    with pytest.raises(RuntimeError, match="size/shape"):  # the byte count no longer matches the element size of the declared dtype
        capi.Model(rebuild(h), cfg)
    h = json.loads(json.dumps(hdr)); h[name]["data_offsets"] = [0, 10 ** 12]
    with pytest.raises(RuntimeError, match="offsets"):
        capi.Model(rebuild(h), cfg)
    h = json.loads(json.dumps(hdr)); h[name]["shape"] = [3, 5]
    with pytest.raises(RuntimeError, match="size/shape"):
        capi.Model(rebuild(h), cfg)


def test_unreadable_vocab(image, tmp_path):
    cfg, img = image
    p = str(tmp_path / "m.safetensors")
    img.tofile(p)
    with pytest.raises(RuntimeError, match="vocab"):
        capi.Model(p, cfg, vocab_path=str(tmp_path / "nope.txt"))


def test_hostile_headers_are_refused_not_crashed(image):
    """Stack-deep __metadata__ nesting, an integer with hundreds of digits, a \\u escape cut off by the end of the header, negative
    and overflowing extents: all answered with an error."""
    cfg, img = image
    hlen = struct.unpack("<Q", img[:8].tobytes())[0]
    hdr = img[8:8 + hlen].tobytes().decode()
    body = img[8 + hlen:].tobytes()

    def with_header(text):
        hb = text.encode()
        return np.frombuffer(struct.pack("<Q", len(hb)) + hb + body, np.uint8)

    deep = '{"__metadata__":' + "[" * 200000 + "]" * 200000 + "," + hdr[1:]
    with pytest.raises(RuntimeError, match="nesting"):
        capi.Model(with_header(deep), cfg)
    name = "encoder_.layers_.0.ffn1_.fc1_.weight"
    h = json.loads(hdr)
    long_int = json.dumps(h).replace('"shape": [256, 128]', '"shape": [' + "9" * 400 + ", 128]", 1)
    assert "9" * 400 in long_int
    with pytest.raises(RuntimeError, match="integer too long"):
        capi.Model(with_header(long_int), cfg)
    with pytest.raises(RuntimeError):
        capi.Model(with_header('{"a\\u12'), cfg)
    h2 = json.loads(hdr); h2[name]["shape"] = [-256, -128]
    with pytest.raises(RuntimeError, match="negative"):
        capi.Model(with_header(json.dumps(h2)), cfg)
    h3 = json.loads(hdr); h3[name]["shape"] = [2 ** 40, 2 ** 40]
    with pytest.raises(RuntimeError, match="exceeds"):
        capi.Model(with_header(json.dumps(h3)), cfg)


def test_config_bounds_are_checked_before_use(image):
    """Non-positive sizes used to reach a modulo / a vector resize (SIGFPE, bad_alloc); now PK_ERR_INVALID."""
    import dataclasses
    cfg, img = image
    for field, val in (("subsampling_channels", 0), ("mel_bins", 0), ("num_layers", -3), ("ffn_intermediate", 0), ("vocab_size", -5),
                       ("max_symbols_per_step", 0), ("hidden_size", 0), ("num_heads", 0)):
        with pytest.raises(RuntimeError):
            capi.Model(img, dataclasses.replace(cfg, **{field: val}))


def test_text_entry_points_refuse_null_arguments():
    import ctypes as C
    L = capi.lib()
    assert L.pk_detokenize(None, None, 0, None, 0) == -1
    assert L.pk_tokenize(None, b"x", None, 0) == -1
    L.pk_group_timestamps.restype = C.c_int
    assert L.pk_group_timestamps(None, None, None, None, None, 0, 0, None, 0, None, None, None, 0) == -1
