"""ctypes binding of include/parakeet_amd.h (libparakeet_amd.so).  Plumbing for tests and bench.py.
Fails loudly if the HIP library is missing -- there is no Python / CPU fallback."""
import ctypes as C
import os

import numpy as np

from .config import ModelConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PK_LIB") or os.path.join(_HERE, "libparakeet_amd.so")   # PK_LIB: experiment builds of the same library
_LIB = None

f32p, i32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class PkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pk_status {code}: {msg}")
        self.code = code


class PkConfig(C.Structure):
    _fields_ = [("mel_bins", C.c_int32), ("subsampling_channels", C.c_int32), ("hidden_size", C.c_int32),
                ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("ffn_intermediate", C.c_int32),
                ("conv_kernel_size", C.c_int32), ("vocab_size", C.c_int32), ("pred_hidden", C.c_int32),
                ("num_lstm_layers", C.c_int32), ("joint_hidden", C.c_int32), ("num_durations", C.c_int32),
                ("durations", C.c_int32 * 8), ("ctc_vocab_size", C.c_int32), ("blank_id", C.c_int32),
                ("max_symbols_per_step", C.c_int32), ("joint_pred_bias", C.c_int32), ("rnnt_head", C.c_int32), ("stft_window_centered", C.c_int32), ("gemm_bf16", C.c_int32),
                ("joint_prefix", C.c_char * 32), ("xscaling", C.c_int32), ("mel_normalize_off", C.c_int32),
                ("encoder_prefix", C.c_char * 32)]


class PkTransformerConfig(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("ffn_intermediate", C.c_int32),
                ("pre_ln", C.c_int32), ("has_final_norm", C.c_int32), ("layer_norm_eps", C.c_float)]


class PkSortformerConfig(C.Structure):
    _fields_ = [("nest", PkConfig), ("transformer", PkTransformerConfig), ("max_speakers", C.c_int32), ("activity_threshold", C.c_float),
                ("att_context_left", C.c_int32), ("att_context_right", C.c_int32)]


class PkKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int32), ("total_ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


class PkOptions(C.Structure):
    _fields_ = [("decoder", C.c_int32), ("timestamps", C.c_int32), ("boost_phrases", C.POINTER(C.c_char_p)),
                ("n_boost_phrases", C.c_int32), ("boost_score", C.c_float)]


class PkWord(C.Structure):
    _fields_ = [("word", C.c_char_p), ("start", C.c_float), ("end", C.c_float), ("confidence", C.c_float)]


class PkResult(C.Structure):
    _fields_ = [("text", C.c_char_p), ("n_tokens", C.c_int32), ("token_ids", i32p), ("start_frame", i32p),
                ("end_frame", i32p), ("confidence", f32p), ("n_words", C.c_int32), ("words", C.POINTER(PkWord))]


def to_pk_config(cfg: ModelConfig) -> PkConfig:
    c = PkConfig()
    c.mel_bins, c.subsampling_channels, c.hidden_size = cfg.mel_bins, cfg.subsampling_channels, cfg.hidden_size
    c.num_layers, c.num_heads, c.ffn_intermediate = cfg.num_layers, cfg.num_heads, cfg.ffn_intermediate
    c.conv_kernel_size, c.vocab_size, c.pred_hidden = cfg.conv_kernel_size, cfg.vocab_size, cfg.pred_hidden
    c.num_lstm_layers, c.joint_hidden, c.num_durations = cfg.num_lstm_layers, cfg.joint_hidden, len(cfg.durations)
    for i, d in enumerate(cfg.durations):
        c.durations[i] = d
    c.ctc_vocab_size, c.blank_id, c.max_symbols_per_step = cfg.ctc_vocab_size, cfg.blank_id, cfg.max_symbols_per_step
    c.joint_pred_bias, c.rnnt_head = 0, int(cfg.head == "rnnt")
    c.stft_window_centered = int(getattr(cfg, "stft_window_centered", False))
    c.gemm_bf16 = int(getattr(cfg, "gemm_bf16", False))
    c.joint_prefix = cfg.joint_prefix.encode()
    c.xscaling = int(getattr(cfg, "xscaling", False))
    c.mel_normalize_off = int(not getattr(cfg, "mel_normalize", True))
    ep = getattr(cfg, "encoder_prefix", "encoder_.")
    c.encoder_prefix = b"" if ep == "encoder_." else ep.encode()
    return c


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.pk_version.restype = C.c_char_p
    L.pk_last_error.restype = C.c_size_t
    L.pk_last_error.argtypes = [C.c_char_p, C.c_size_t]
    L.pk_config_preset.argtypes = [C.c_char_p, C.POINTER(PkConfig)]
    L.pk_model_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(PkConfig), C.POINTER(C.c_void_p)]
    L.pk_model_to_gpu.argtypes = [C.c_void_p, C.c_int]
    L.pk_model_free.argtypes = [C.c_void_p]
    L.pk_model_config.argtypes = [C.c_void_p, C.POINTER(PkConfig)]
    L.pk_mel.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int64, f32p, f32p]
    L.pk_mel_num_frames.argtypes = [C.c_int64]
    L.pk_encoder_num_frames.argtypes = [C.c_int]
    L.pk_diag_math.argtypes = [C.c_int, f32p, f32p, C.c_int64]
    L.pk_diag_math_exhaustive.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pk_diag_gemm.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, C.c_int, f32p, C.c_float, f32p]
    L.pk_diag_gemm_bf16.argtypes = L.pk_diag_gemm.argtypes
    L.pk_diag_gemm_bf16_a16.argtypes = L.pk_diag_gemm.argtypes
    L.pk_diag_layernorm.argtypes = [f32p, C.c_int64, C.c_int, f32p, f32p, C.c_float, f32p]
    L.pk_diag_ln_gemm.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_float, f32p, f32p, C.c_int, C.c_int, f32p, f32p]
    L.pk_diag_sum64.argtypes = [f32p, C.c_int, C.c_int, f32p]
    for name, at in _LATE_SIGNATURES.items():
        if hasattr(L, name):
            getattr(L, name).argtypes = at
    _LIB = L
    return L


_LATE_SIGNATURES = {
    "pk_encode": [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p],
    "pk_conformer_blocks": [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p],
    "pk_subsample": [C.c_void_p, f32p, C.c_int, C.c_int, f32p],
    "pk_ctc_decode": [C.c_void_p, f32p, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f32p, f32p],
    "pk_tdt_decode": [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f32p, i32p],
    "pk_model_set_decode_loop": [C.c_void_p, C.c_int],
    "pk_plan_batches": [i64p, C.c_int, i32p, i32p, C.POINTER(C.c_int)],
    "pk_ragged_extents": [i64p, C.c_int, i32p, i32p, i64p],
    "pk_tdt_score": [C.c_void_p, f32p, C.c_int, i32p, i32p, C.c_int, f32p, f32p, C.POINTER(C.c_int)],
    "pk_mel_ragged": [C.c_void_p, f32p, i64p, C.c_int, f32p, f32p],
    "pk_encode_ragged": [C.c_void_p, f32p, i32p, C.c_int, C.c_int, C.c_int, f32p],
    "pk_conformer_blocks_ragged": [C.c_void_p, f32p, i32p, C.c_int, C.c_int, C.c_int, f32p],
    "pk_ctc_decode_ragged": [C.c_void_p, f32p, i32p, C.c_int, i32p, i32p, i32p, i32p, f32p, f32p],
    "pk_tdt_decode_ragged": [C.c_void_p, f32p, i32p, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f32p, i32p],
    "pk_batch_create_ragged": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)],
    "pk_batch_upload_ragged": [C.c_void_p, f32p, i64p, C.c_int],
    "pk_batch_upload_ragged_async": [C.c_void_p, f32p, i64p, C.c_int],
    "pk_batch_create": [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_void_p)],
    "pk_batch_free": [C.c_void_p],
    "pk_batch_upload": [C.c_void_p, f32p, C.c_int],
    "pk_batch_run": [C.c_void_p, C.c_int],
    "pk_batch_upload_async": [C.c_void_p, f32p, C.c_int],
    "pk_batch_results_done": [C.c_void_p, C.POINTER(C.c_int), i32p, i32p, i32p, i32p, f32p],
    "pk_batch_results_back": [C.c_void_p, C.c_int, C.POINTER(C.c_int), i32p, i32p, i32p, i32p, f32p],
    "pk_batch_results_available": [C.c_void_p],
    "pk_batch_set_decode_group": [C.c_void_p, C.c_int],
    "pk_batch_sync": [C.c_void_p],
    "pk_batch_set_decode_overlap": [C.c_void_p, C.c_int],
    "pk_batch_margins": [C.c_void_p, C.c_int, f32p],
    "pk_decode_margins": [C.c_void_p, f32p, C.c_int],
    "pk_batch_max_tokens": [C.c_void_p],
    "pk_batch_results": [C.c_void_p, i32p, i32p, i32p, i32p, f32p],
    "pk_batch_run_timed": [C.c_void_p, C.c_int, f32p],
    "pk_batch_profile": [C.c_void_p, C.c_int, C.POINTER(PkKernelStat), C.c_int],
    "pk_transcribe_pcm": [C.c_void_p, f32p, i64p, C.c_int, C.POINTER(PkOptions), C.POINTER(C.POINTER(PkResult))],
    "pk_results_free": [C.POINTER(PkResult), C.c_int],
    "pk_read_wav": [C.c_char_p, C.POINTER(f32p), i64p, C.POINTER(C.c_int)],
    "pk_free": [C.c_void_p],
    "pk_vocab_size": [C.c_void_p],
    "pk_detokenize": [C.c_void_p, i32p, C.c_int, C.c_char_p, C.c_int],
    "pk_tokenize": [C.c_void_p, C.c_char_p, i32p, C.c_int],
    "pk_set_boost_tokens": [C.c_void_p, i32p, i32p, C.c_int, C.c_float],
    "pk_set_boost_phrases": [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_float],
    "pk_boost_trie_size": [C.c_void_p],
    "pk_group_timestamps": [C.c_void_p, i32p, i32p, i32p, f32p, C.c_int, C.c_int, C.c_char_p, C.c_int, f32p, f32p, f32p, C.c_int],
    "pk_group_create": [C.c_char_p, C.c_char_p, C.POINTER(PkConfig), i32p, C.c_int, C.POINTER(C.c_void_p)],
    "pk_group_free": [C.c_void_p],
    "pk_group_size": [C.c_void_p],
    "pk_group_transcribe_pcm": [C.c_void_p, f32p, i64p, C.c_int, C.POINTER(PkOptions), C.POINTER(C.POINTER(PkResult))],
    "pk_group_last_stats": [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), i32p],
    "pk_group_verify_exchange": [C.c_void_p, C.POINTER(PkResult), C.c_int, C.POINTER(C.c_int)],
}


def _f(a):
    return a.ctypes.data_as(f32p)


def _i(a):
    return a.ctypes.data_as(i32p)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def check(st):
    if st != 0:
        buf = C.create_string_buffer(2048)
        lib().pk_last_error(buf, 2048)
        raise PkError(st, buf.value.decode(errors="replace"))


def plan_batches(n_samples):
    """pk_plan_batches: how the one-call API packs clips of these lengths -> (batch_of_clip, pos_in_batch, n_batches).  Host logic."""
    n = np.ascontiguousarray(n_samples, np.int64)
    b = np.zeros(len(n), np.int32); p = np.zeros(len(n), np.int32); nb = C.c_int(0)
    check(lib().pk_plan_batches(n.ctypes.data_as(i64p), len(n), _i(b), _i(p), C.byref(nb)))
    return b, p, nb.value


def ragged_extents(n_samples):
    n = np.ascontiguousarray(n_samples, np.int64)
    tm = np.zeros(len(n), np.int32); t = np.zeros(len(n), np.int32); tot = np.zeros(7, np.int64)
    check(lib().pk_ragged_extents(n.ctypes.data_as(i64p), len(n), _i(tm), _i(t), tot.ctypes.data_as(i64p)))
    return tm, t, tot


def device_count():
    return lib().pk_device_count()


# ---- diagnostics ---------------------------------------------------------------------------------
MATH_FN = {"exp": 0, "log": 1, "tanh": 2, "sigmoid": 3, "silu": 4, "sqrt": 5, "rcp": 6, "sigmoid4": 8, "silu4": 9}
EPI = {"none": 0, "relu": 1, "silu": 2, "resid": 3, "glu": 4}


def diag_math(fn, x):
    x = _c(x)
    y = np.empty_like(x)
    check(lib().pk_diag_math(MATH_FN[fn], _f(x), _f(y), x.size))
    return y


def diag_math_exhaustive(fn, guarded=False):
    """all 2^32 bit patterns through the device-side identity of pk_diag_math_exhaustive -> (checked, mismatches, first_bad)"""
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    check(lib().pk_diag_math_exhaustive(MATH_FN[fn] + (10 if guarded else 0), C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def resample(pcm, src_rate, dst_rate):
    """pk_resample: the reference's Kaiser-windowed sinc resampler (src/audio_io.cpp:123-195)."""
    pcm = _c(pcm)
    L = lib()
    L.pk_resample.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, C.POINTER(f32p), C.POINTER(C.c_int64)]
    L.pk_free.argtypes = [C.c_void_p]
    out, n = f32p(), C.c_int64(0)
    check(L.pk_resample(_f(pcm), pcm.size, src_rate, dst_rate, C.byref(out), C.byref(n)))
    r = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
    L.pk_free(out)
    return r


def read_audio(path, target_rate=16000):
    """pk_read_audio: WAV -> mono -> resampled to target_rate; returns (pcm, original_rate)."""
    L = lib()
    L.pk_read_audio.argtypes = [C.c_char_p, C.c_int, C.POINTER(f32p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.pk_free.argtypes = [C.c_void_p]
    out, n, sr = f32p(), C.c_int64(0), C.c_int(0)
    check(L.pk_read_audio(path.encode(), target_rate, C.byref(out), C.byref(n), C.byref(sr)))
    r = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
    L.pk_free(out)
    return r, sr.value


def diag_gemm(A, W, bias=None, epi="none", resid=None, alpha=1.0, bf16=False, a16=False):
    A, W = _c(A), _c(W)
    M, K = A.shape
    N = W.shape[0] // 2 if epi == "glu" else W.shape[0]
    b = _c(bias) if bias is not None else None
    r = _c(resid) if resid is not None else None
    out = np.empty((M, N), np.float32)
    fn = (lib().pk_diag_gemm_bf16_a16 if a16 else lib().pk_diag_gemm_bf16) if bf16 else lib().pk_diag_gemm
    check(fn(M, N, K, _f(A), _f(W), _f(b) if b is not None else None, EPI[epi], _f(r) if r is not None else None, alpha, _f(out)))
    return out


def diag_ln_gemm_bf16(A, gamma, beta, W, bias=None, epi="none", resid=None, alpha=1.0, eps=1e-5):
    """pk_diag_ln_gemm_bf16: epi(bf16(LayerNorm(A)) bf16(W)^T + bias) on the small-M bf16 kernel with the LayerNorm folded in."""
    A, W, gamma, beta = _c(A), _c(W), _c(gamma), _c(beta)
    M, K = A.shape
    N = W.shape[0] // 2 if epi == "glu" else W.shape[0]
    b = _c(bias) if bias is not None else None
    r = _c(resid) if resid is not None else None
    out = np.empty((M, N), np.float32)
    L = lib()
    L.pk_diag_ln_gemm_bf16.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, C.c_float, f32p, f32p, C.c_int, f32p, C.c_float, f32p]
    check(L.pk_diag_ln_gemm_bf16(M, N, K, _f(A), _f(gamma), _f(beta), eps, _f(W), _f(b) if b is not None else None, EPI[epi],
                                 _f(r) if r is not None else None, alpha, _f(out)))
    return out


def diag_ln2_gemm_bf16(A, pre_gamma, pre_beta, gamma, beta, W, bias, eps=1e-5):
    """pk_diag_ln2_gemm_bf16: silu(bf16(LN(LN(A; pre); gamma, beta)) bf16(W)^T + bias) and LN(A; pre) on the small-M bf16 kernel."""
    A, W, pg, pb, g, b, bias = (_c(v) for v in (A, W, pre_gamma, pre_beta, gamma, beta, bias))
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), np.float32); pre = np.empty((M, K), np.float32)
    L = lib()
    L.pk_diag_ln2_gemm_bf16.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_float, f32p, f32p, f32p, f32p]
    check(L.pk_diag_ln2_gemm_bf16(M, N, K, _f(A), _f(pg), _f(pb), _f(g), _f(b), eps, _f(W), _f(bias), _f(out), _f(pre)))
    return out, pre


def diag_smallm_bf16_tiles(on):
    """pk_diag_smallm_bf16_tiles: the bf16 diag products with / without the operand-tiled weight copy (test switch, process-wide)."""
    L = lib()
    L.pk_diag_smallm_bf16_tiles.argtypes = [C.c_int]
    check(L.pk_diag_smallm_bf16_tiles(int(on)))


def diag_ffn_bf16_smallm(x, gamma, beta, W1, b1, W2, b2, act_tiles, eps=1e-5):
    """pk_diag_ffn_bf16_smallm: x + 0.5 * ffn(LN(x)) of a streaming chunk on the small-M bf16 kernel; act_tiles = fc1 activations in 8-row operand tiles."""
    x, gamma, beta, W1, b1, W2, b2 = (_c(v) for v in (x, gamma, beta, W1, b1, W2, b2))
    M, d = x.shape
    f = W1.shape[0]
    out = np.empty((M, d), np.float32)
    L = lib()
    L.pk_diag_ffn_bf16_smallm.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, C.c_float, f32p, f32p, f32p, f32p, C.c_int, f32p]
    check(L.pk_diag_ffn_bf16_smallm(M, d, f, _f(x), _f(gamma), _f(beta), eps, _f(W1), _f(b1), _f(W2), _f(b2), int(act_tiles), _f(out)))
    return out


def diag_glu_dwconv_bf16(A, W, bias, cache_in, has_cache, dw_w, dw_bias, bn_mean, bn_rstd, bn_g, bn_b, c, fused, gamma=None, beta=None, eps=1e-5):
    """pk_diag_glu_dwconv_bf16: pw1 (GLU) + causal depthwise conv + BatchNorm + SiLU of a streaming chunk; fused = the conv in the product's epilogue."""
    A, W, bias, cache_in = _c(A), _c(W), _c(bias), _c(cache_in)
    M, d = A.shape
    S = M // c
    ps = [_c(v) for v in (dw_w, dw_bias, bn_mean, bn_rstd, bn_g, bn_b)]
    g, b = (_c(gamma), _c(beta)) if gamma is not None else (None, None)
    out = np.empty((M, d), np.float32); cache_out = np.empty((S, 8, d), np.float32)
    L = lib()
    L.pk_diag_glu_dwconv_bf16.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, C.c_float, f32p, f32p, f32p, C.c_int] + [f32p] * 6 + [C.c_int, f32p, f32p]
    check(L.pk_diag_glu_dwconv_bf16(S, c, d, _f(A), _f(g) if g is not None else None, _f(b) if b is not None else None, eps, _f(W), _f(bias), _f(cache_in),
                                    int(has_cache), *[_f(v) for v in ps], int(fused), _f(out), _f(cache_out)))
    return out, cache_out


class Stream:
    """pk_stream_*: n lock-step streaming sessions on the GPU (reference NemotronTranscriber::transcribe_chunk)."""

    def __init__(self, model, n_streams, att_context_left=70, att_context_right=0):
        L = lib()
        L.pk_stream_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pk_stream_free.argtypes = [C.c_void_p]
        L.pk_stream_free.restype = None
        L.pk_stream_reset.argtypes = [C.c_void_p]
        L.pk_stream_push.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f32p]
        L.pk_stream_mel.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_int, C.POINTER(C.c_int)]
        L.pk_stream_encode.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_int, C.POINTER(C.c_int)]
        L.pk_stream_decode.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f32p]
        self.model, self.S = model, n_streams
        self._h = C.c_void_p()
        check(L.pk_stream_create(model._h, n_streams, att_context_left, att_context_right, C.byref(self._h)))

    def _tok(self, fn, x, n, mt):
        ids = np.zeros((self.S, mt), np.int32); st = np.zeros((self.S, mt), np.int32); en = np.zeros((self.S, mt), np.int32)
        cf = np.zeros((self.S, mt), np.float32); lens = np.zeros(self.S, np.int32)
        check(fn(self._h, _f(x), n, mt, _i(ids), _i(lens), _i(st), _i(en), _f(cf)))
        return dict(ids=ids, lens=lens, start=st, end=en, conf=cf)

    def push(self, pcm, max_tokens=64):
        pcm = _c(pcm)
        assert pcm.shape[0] == self.S
        return self._tok(lib().pk_stream_push, pcm, pcm.shape[1], max_tokens)

    def mel(self, pcm):
        pcm = _c(pcm)
        cap = pcm.shape[1] // 160 + 8
        out = np.zeros((self.S, cap, self.model.cfg.mel_bins), np.float32)
        n = C.c_int(0)
        check(lib().pk_stream_mel(self._h, _f(pcm), pcm.shape[1], _f(out), cap, C.byref(n)))
        return np.ascontiguousarray(out.reshape(-1)[: self.S * n.value * self.model.cfg.mel_bins].reshape(self.S, n.value, -1))

    def encode(self, mel):
        mel = _c(mel)
        cap = mel.shape[1] // 8 + 2
        out = np.zeros((self.S, cap, self.model.cfg.hidden_size), np.float32)
        n = C.c_int(0)
        check(lib().pk_stream_encode(self._h, _f(mel), mel.shape[1], _f(out), cap, C.byref(n)))
        d = self.model.cfg.hidden_size
        return np.ascontiguousarray(out.reshape(-1)[: self.S * n.value * d].reshape(self.S, n.value, d))

    def decode(self, enc, max_tokens=64):
        enc = _c(enc)
        return self._tok(lib().pk_stream_decode, enc, enc.shape[1], max_tokens)

    def score(self, enc, labels, dur_idx, n_steps, rows=True):
        """pk_stream_score: every stream walks its GIVEN decisions labels[s][:n_steps[s]] / dur_idx[s][...] on this chunk's enc[S][c][d];
        -> label_lp [S][cap][V] (rows=True), dur_lp [S][cap][D], n [S] steps walked."""
        enc = _c(enc)
        labels = _c(labels, np.int32); dur_idx = _c(dur_idx, np.int32); n_steps = _c(n_steps, np.int32)
        assert enc.shape[0] == self.S and labels.shape == dur_idx.shape and labels.shape[0] == self.S and n_steps.shape == (self.S,)
        cap = labels.shape[1]
        cfg = self.model.cfg
        V, D = cfg.vocab_size, len(cfg.durations)
        L = lib()
        L.pk_stream_score.argtypes = [C.c_void_p, f32p, C.c_int, i32p, i32p, i32p, C.c_int, f32p, f32p, i32p]
        llp = np.zeros((self.S, cap, V), np.float32) if rows else None
        dlp = np.zeros((self.S, cap, D), np.float32)
        nd = np.zeros(self.S, np.int32)
        check(L.pk_stream_score(self._h, _f(enc), enc.shape[1], _i(labels), _i(dur_idx), _i(n_steps), cap, _f(llp) if rows else None, _f(dlp), _i(nd)))
        return dict(label_lp=llp, dur_lp=dlp, n=nd)

    def reset(self):
        check(lib().pk_stream_reset(self._h))

    def close(self):
        if self._h:
            lib().pk_stream_free(self._h)
            self._h = None


class Transformer:
    """pk_transformer_*: TransformerEncoder of the reference (src/transformer.cpp) on the GPU."""

    def __init__(self, weights_path, prefix, hidden_size, num_layers, num_heads, ffn_intermediate, pre_ln=True, has_final_norm=False,
                 layer_norm_eps=1e-5, device=0):
        L = lib()
        L.pk_transformer_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(PkTransformerConfig), C.c_int, C.POINTER(C.c_void_p)]
        L.pk_transformer_forward.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, f32p]
        L.pk_transformer_free.argtypes = [C.c_void_p]
        L.pk_transformer_free.restype = None
        c = PkTransformerConfig(hidden_size, num_layers, num_heads, ffn_intermediate, int(pre_ln), int(has_final_norm), layer_norm_eps)
        self._h = C.c_void_p()
        check(L.pk_transformer_load(weights_path.encode(), prefix.encode(), C.byref(c), device, C.byref(self._h)))

    def forward(self, x):
        x = _c(x)
        B, T, _ = x.shape
        y = np.empty_like(x)
        check(lib().pk_transformer_forward(self._h, _f(x), B, T, _f(y)))
        return y

    def close(self):
        if self._h:
            lib().pk_transformer_free(self._h)
            self._h = None


class Sortformer:
    """pk_sortformer_*: Sortformer diarization (reference include/parakeet/sortformer.hpp:99-129) on the GPU."""

    def __init__(self, weights_path, sf, device=0):
        L = lib()
        L.pk_sortformer_load.argtypes = [C.c_char_p, C.POINTER(PkSortformerConfig), C.c_int, C.POINTER(C.c_void_p)]
        L.pk_sortformer_free.argtypes = [C.c_void_p]
        L.pk_sortformer_free.restype = None
        L.pk_sortformer_forward.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, f32p, C.POINTER(C.c_int)]
        L.pk_sortformer_forward_pcm.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int64, f32p, C.POINTER(C.c_int)]
        L.pk_sortformer_segments.argtypes = [f32p, C.c_int, C.c_int, C.c_float, i32p, f32p, f32p, C.c_int]
        self.sf = sf
        c = PkSortformerConfig()
        c.nest = to_pk_config(sf.nest_encoder)
        c.transformer = PkTransformerConfig(sf.transformer_hidden, sf.transformer_layers, sf.transformer_heads, sf.transformer_ffn,
                                            int(sf.pre_ln), int(sf.has_final_norm), 1e-5)
        c.max_speakers, c.activity_threshold = sf.max_speakers, sf.activity_threshold
        c.att_context_left, c.att_context_right = sf.att_context_left, sf.att_context_right
        L.pk_sortformer_diarize_chunk.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_int, C.POINTER(C.c_int)]
        L.pk_sortformer_stream_reset.argtypes = [C.c_void_p]
        self._h = C.c_void_p()
        check(L.pk_sortformer_load(weights_path.encode(), C.byref(c), device, C.byref(self._h)))

    def forward(self, feats):
        """Sortformer::forward: feats [B][Tm][mel] -> probs [B][T][S]."""
        feats = _c(feats)
        B, Tm, _ = feats.shape
        T = lib().pk_encoder_num_frames(Tm)
        probs = np.zeros((B, T, self.sf.max_speakers), np.float32)
        check(lib().pk_sortformer_forward(self._h, _f(feats), B, Tm, _f(probs), None))
        return probs

    def diarize_chunk(self, feats):
        """Sortformer::diarize_chunk on feats [n_frames][mel]: -> probs [c][S] of this chunk (c may be 0)."""
        feats = _c(feats)
        cap = feats.shape[0] // 8 + 4
        probs = np.zeros((cap, self.sf.max_speakers), np.float32)
        n = C.c_int(0)
        check(lib().pk_sortformer_diarize_chunk(self._h, _f(feats), feats.shape[0], _f(probs), cap, C.byref(n)))
        return probs[:n.value]

    def reset_stream(self):
        check(lib().pk_sortformer_stream_reset(self._h))

    def forward_pcm(self, pcm):
        pcm = _c(pcm)
        if pcm.ndim == 1:
            pcm = pcm[None]
        B, n = pcm.shape
        T = lib().pk_encoder_num_frames(lib().pk_mel_num_frames(n))
        probs = np.zeros((B, T, self.sf.max_speakers), np.float32)
        check(lib().pk_sortformer_forward_pcm(self._h, _f(pcm), B, n, _f(probs), None))
        return probs

    def close(self):
        if self._h:
            lib().pk_sortformer_free(self._h)
            self._h = None


def sortformer_segments(probs, threshold=0.5):
    """Sortformer::probs_to_segments on probs [T][S] -> list of (speaker, start_s, end_s)."""
    L = lib()
    L.pk_sortformer_segments.argtypes = [f32p, C.c_int, C.c_int, C.c_float, i32p, f32p, f32p, C.c_int]
    probs = _c(probs)
    T, S = probs.shape
    cap = S * (T // 2 + 2)
    spk = np.zeros(cap, np.int32); a = np.zeros(cap, np.float32); b = np.zeros(cap, np.float32)
    n = L.pk_sortformer_segments(_f(probs), T, S, threshold, _i(spk), _f(a), _f(b), cap)
    return [(int(spk[i]), float(a[i]), float(b[i])) for i in range(n)]


class Frontend:
    """pk_frontend_*: preprocess_audio without a model (reference src/audio.cpp:100-158)."""

    def __init__(self, n_mels=80, normalize=True, window_centered=False, device=0):
        L = lib()
        L.pk_frontend_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pk_frontend_features.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, C.POINTER(C.c_int)]
        L.pk_frontend_free.argtypes = [C.c_void_p]
        L.pk_frontend_free.restype = None
        self.n_mels = n_mels
        self._h = C.c_void_p()
        check(L.pk_frontend_create(n_mels, int(normalize), int(window_centered), device, C.byref(self._h)))

    def features(self, pcm):
        pcm = _c(pcm).ravel()
        nf = lib().pk_mel_num_frames(pcm.size)
        out = np.zeros((nf, self.n_mels), np.float32)
        check(lib().pk_frontend_features(self._h, _f(pcm), pcm.size, _f(out), None))
        return out

    def close(self):
        if self._h:
            lib().pk_frontend_free(self._h)
            self._h = None


def read_audio_memory(data: bytes, target_rate=16000):
    """read_audio(const uint8_t*, size_t, target) -> (mono pcm at target_rate, original rate, channels)."""
    L = lib()
    L.pk_read_audio_memory.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(f32p), i64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    p = f32p(); n = C.c_int64(0); sr = C.c_int(0); ch = C.c_int(0)
    check(L.pk_read_audio_memory(data, len(data), target_rate, C.byref(p), C.byref(n), C.byref(sr), C.byref(ch)))
    out = np.ctypeslib.as_array(p, shape=(max(1, n.value),))[:n.value].copy()
    L.pk_free(p)
    return out, sr.value, ch.value


def audio_info(path):
    """get_audio_duration's header walk -> (sample_rate, channels, frames)."""
    L = lib()
    L.pk_audio_info.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), i64p]
    sr = C.c_int(0); ch = C.c_int(0); n = C.c_int64(0)
    check(L.pk_audio_info(path.encode(), C.byref(sr), C.byref(ch), C.byref(n)))
    return sr.value, ch.value, n.value


def diag_layernorm(x, g, b, eps=1e-5):
    x, g, b = _c(x), _c(g), _c(b)
    y = np.empty_like(x)
    check(lib().pk_diag_layernorm(_f(x), x.size // x.shape[-1], x.shape[-1], _f(g), _f(b), eps, _f(y)))
    return y


def diag_ln_gemm(A, gamma, beta, W, bias=None, epi="none", fold=True, pre_gamma=None, pre_beta=None, eps=1e-5):
    """out = epi(LN(X) W^T + bias) on a large fp32 batch, X = A or LN(A; pre_gamma, pre_beta); fold: the statistics pass + the tile GEMM that normalises
    while staging A (what the engine runs), else LayerNorm launch + GEMM.  Returns (out, X or None)."""
    A, W, gamma, beta = _c(A), _c(W), _c(gamma), _c(beta)
    M, K = A.shape
    e = EPI[epi]
    N = W.shape[0] // 2 if epi == "glu" else W.shape[0]
    out = np.empty((M, N), np.float32)
    y1 = np.empty((M, K), np.float32) if pre_gamma is not None else None
    pg = _c(pre_gamma) if pre_gamma is not None else None
    pb = _c(pre_beta) if pre_beta is not None else None
    bb = _c(bias) if bias is not None else None
    check(lib().pk_diag_ln_gemm(M, N, K, _f(A), _f(pg) if pg is not None else None, _f(pb) if pb is not None else None, _f(gamma), _f(beta), eps, _f(W),
                                _f(bb) if bb is not None else None, e, 1 if fold else 0, _f(out), _f(y1) if y1 is not None else None))
    return out, y1


def diag_sum64(x):
    x = _c(x)
    out = np.empty(x.shape[0], np.float32)
    check(lib().pk_diag_sum64(_f(x), x.shape[0], x.shape[1], _f(out)))
    return out


class Batch:
    """pk_batch: the resident pipeline (clips of one length stay in HBM; decode(k) overlaps encoder(k+1))."""

    def __init__(self, model, max_clips, n_samples):
        self._h = C.c_void_p()
        self.n_clips = 0
        self._cap = max_clips
        check(lib().pk_batch_create(model._h, max_clips, n_samples, C.byref(self._h)))

    @classmethod
    def ragged(cls, model, max_clips, max_total_samples, max_clip_samples):
        """pk_batch_create_ragged: a pipeline whose runs take clips of ANY lengths (packed, no padding)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.n_clips = 0
        self._cap = max_clips
        check(lib().pk_batch_create_ragged(model._h, max_clips, int(max_total_samples), int(max_clip_samples), C.byref(self._h)))
        return self

    @staticmethod
    def _pack(clips):
        clips = [_c(c).ravel() for c in clips]
        off = np.zeros(len(clips) + 1, np.int64)
        off[1:] = np.cumsum([len(c) for c in clips])
        return np.concatenate(clips), off

    def upload_ragged(self, clips):
        pcm, off = self._pack(clips)
        self.n_clips = len(off) - 1
        check(lib().pk_batch_upload_ragged(self._h, _f(pcm), off.ctypes.data_as(i64p), self.n_clips))

    def upload_ragged_async(self, clips):
        pcm, off = self._pack(clips)
        self._staged = pcm
        self._staged_clips = len(off) - 1
        check(lib().pk_batch_upload_ragged_async(self._h, _f(pcm), off.ctypes.data_as(i64p), len(off) - 1))

    def upload(self, pcm):
        pcm = _c(pcm)
        self.n_clips = pcm.shape[0]
        check(lib().pk_batch_upload(self._h, _f(pcm), self.n_clips))

    def upload_async(self, pcm):
        """Stage the NEXT batch under the running encoder (double-buffered PCM, no flush)."""
        pcm = _c(pcm)
        self._staged = pcm                                   # keep the host array alive until the copy has been consumed
        self._staged_clips = pcm.shape[0]
        check(lib().pk_batch_upload_async(self._h, _f(pcm), pcm.shape[0]))

    def run(self, decoder="tdt"):
        if getattr(self, "_staged_clips", 0):
            self.n_clips, self._staged_clips = self._staged_clips, 0
        check(lib().pk_batch_run(self._h, {"ctc": 0, "tdt": 1}[decoder]))

    def set_decode_group(self, group):
        """Throughput mode: the TDT loops of `group` consecutive runs are decoded as one lock-step batch (results unchanged, later)."""
        check(lib().pk_batch_set_decode_group(self._h, int(group)))

    def set_decode_overlap(self, on):
        """on: decode on a second stream under the next encoder (default); off: on the encoder's stream, after it.  Same results."""
        check(lib().pk_batch_set_decode_overlap(self._h, int(bool(on))))

    def sync(self):
        check(lib().pk_batch_sync(self._h))

    def results_available(self):
        return int(lib().pk_batch_results_available(self._h))

    def results_back(self, back=0):
        """Results of the (back+1)-th newest run whose decode has finished; no flush."""
        mt = lib().pk_batch_max_tokens(self._h)
        cap = self._cap
        ids = np.zeros((cap, mt), np.int32); st = np.zeros((cap, mt), np.int32); en = np.zeros((cap, mt), np.int32)
        cf = np.zeros((cap, mt), np.float32); lens = np.zeros(cap, np.int32)
        n = C.c_int(0)
        check(lib().pk_batch_results_back(self._h, int(back), C.byref(n), _i(ids), _i(lens), _i(st), _i(en), _f(cf)))
        B = n.value
        return dict(ids=ids[:B], lens=lens[:B], start=st[:B], end=en[:B], conf=cf[:B])

    def margins(self, back=0, n_clips=None):
        """pk_batch_margins: smallest top-1 / top-2 label log-prob margin of every clip of the (back+1)-th newest finished run."""
        mg = np.zeros(self._cap, np.float32)
        check(lib().pk_batch_margins(self._h, int(back), _f(mg)))
        return mg[: (n_clips if n_clips is not None else self._cap)]

    def results_done(self):
        """Results of the newest batch whose decode has finished (run k's decode completes inside run k+1); no flush."""
        mt = lib().pk_batch_max_tokens(self._h)
        cap = self._cap
        ids = np.zeros((cap, mt), np.int32); st = np.zeros((cap, mt), np.int32); en = np.zeros((cap, mt), np.int32)
        cf = np.zeros((cap, mt), np.float32); lens = np.zeros(cap, np.int32)
        n = C.c_int(0)
        check(lib().pk_batch_results_done(self._h, C.byref(n), _i(ids), _i(lens), _i(st), _i(en), _f(cf)))
        B = n.value
        return dict(ids=ids[:B], lens=lens[:B], start=st[:B], end=en[:B], conf=cf[:B])

    def results(self):
        B, mt = self.n_clips, lib().pk_batch_max_tokens(self._h)
        ids = np.zeros((B, mt), np.int32); st = np.zeros((B, mt), np.int32); en = np.zeros((B, mt), np.int32)
        cf = np.zeros((B, mt), np.float32); lens = np.zeros(B, np.int32)
        check(lib().pk_batch_results(self._h, _i(ids), _i(lens), _i(st), _i(en), _f(cf)))
        return dict(ids=ids, lens=lens, start=st, end=en, conf=cf)

    def close(self):
        if self._h:
            lib().pk_batch_free(self._h)
            self._h = None


# ---- model -------------------------------------------------------------------------------------------
def pack_clips(clips):
    """(pcm, offsets) of a list of clips: what pk_transcribe_pcm takes (one copy of the audio; do it outside a timed region)."""
    clips = [_c(c).ravel() for c in clips]
    off = np.zeros(len(clips) + 1, np.int64)
    off[1:] = np.cumsum([len(c) for c in clips])
    return np.concatenate(clips), off


def _transcribe(fn, handle, clips, decoder, timestamps, boost_phrases, boost_score, with_raw=None):
    if isinstance(clips, tuple):                             # already packed: (pcm, offsets)
        pcm, off = clips
        pcm, off = _c(pcm), np.ascontiguousarray(off, np.int64)
        clips = range(len(off) - 1)
    else:
        pcm, off = pack_clips(clips)
    opt = PkOptions()
    opt.decoder = {"ctc": 0, "tdt": 1}[decoder]
    opt.timestamps = 1 if timestamps else 0
    keep = (C.c_char_p * max(1, len(boost_phrases)))(*[p.encode() for p in boost_phrases])
    opt.boost_phrases = keep
    opt.n_boost_phrases = len(boost_phrases)
    opt.boost_score = boost_score
    res = C.POINTER(PkResult)()
    check(fn(handle, _f(pcm), off.ctypes.data_as(i64p), len(clips), C.byref(opt), C.byref(res)))
    out = []
    for i in range(len(clips)):
        r = res[i]
        d = dict(text=(r.text or b"").decode(), token_ids=[r.token_ids[k] for k in range(r.n_tokens)])
        if timestamps:
            d["start"] = [r.start_frame[k] for k in range(r.n_tokens)]
            d["end"] = [r.end_frame[k] for k in range(r.n_tokens)]
            d["conf"] = [r.confidence[k] for k in range(r.n_tokens)]
            d["words"] = [(r.words[k].word.decode(), r.words[k].start, r.words[k].end, r.words[k].confidence) for k in range(r.n_words)]
        out.append(d)
    if with_raw is not None:
        with_raw(res, len(clips))                            # e.g. pk_group_verify_exchange on the pk_result array itself
    lib().pk_results_free(res, len(clips))
    return out


class Group:
    """pk_group: one model replica per GPU of this process, utterance batches dealt round-robin, one host thread and one two-stream
    pipeline per device, no collective (include/parakeet_amd.h, "one node, several GPUs")."""

    def __init__(self, weights_path, cfg: ModelConfig, vocab_path: str = None, devices=None):
        self.cfg = cfg
        self._h = C.c_void_p()
        pc = to_pk_config(cfg)
        dev = np.ascontiguousarray(devices, np.int32) if devices is not None else None
        check(lib().pk_group_create(weights_path.encode(), vocab_path.encode() if vocab_path else None, C.byref(pc),
                                    _i(dev) if dev is not None else None, len(dev) if dev is not None else 0, C.byref(self._h)))

    def size(self):
        return lib().pk_group_size(self._h)

    def transcribe_pcm(self, clips, decoder="tdt", timestamps=False, boost_phrases=(), boost_score=5.0, verify_exchange=False):
        """verify_exchange: also run pk_group_verify_exchange (the RCCL all-reduce + all-gather of the token matrix) on the results;
        self.rccl_ranks then holds the communicator's rank count."""
        raw = None
        if verify_exchange:
            def raw(res, n):
                ranks = C.c_int(0)
                check(lib().pk_group_verify_exchange(self._h, res, n, C.byref(ranks)))
                self.rccl_ranks = ranks.value
        return _transcribe(lib().pk_group_transcribe_pcm, self._h, clips, decoder, timestamps, boost_phrases, boost_score, with_raw=raw)

    def last_stats(self):
        wall, audio = C.c_double(0), C.c_double(0)
        per = np.zeros(self.size(), np.int32)
        check(lib().pk_group_last_stats(self._h, C.byref(wall), C.byref(audio), _i(per)))
        return dict(wall_ms_max=wall.value, audio_seconds=audio.value, clips_per_rank=per.tolist())

    def close(self):
        if getattr(self, "_h", None):
            lib().pk_group_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """Thin handle over pk_model (mirrors parakeet::Transcriber's ctor + to_gpu(), transcribe.hpp:59-71)."""

    def __init__(self, weights_path, cfg: ModelConfig, vocab_path: str = None, device: int = None):
        """weights_path: a safetensors file, or its bytes (bytes / uint8 ndarray), e.g. as received from a broadcast."""
        self.cfg = cfg
        self._h = C.c_void_p()
        pc = to_pk_config(cfg)
        vp = vocab_path.encode() if vocab_path else None
        if isinstance(weights_path, str):
            check(lib().pk_model_load(weights_path.encode(), vp, C.byref(pc), C.byref(self._h)))
        else:
            img = np.frombuffer(weights_path, np.uint8) if isinstance(weights_path, (bytes, bytearray)) else np.ascontiguousarray(weights_path, np.uint8)
            L = lib()
            L.pk_model_load_buffer.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(PkConfig), C.POINTER(C.c_void_p)]
            check(L.pk_model_load_buffer(img.ctypes.data_as(C.c_void_p), img.size, vp, C.byref(pc), C.byref(self._h)))
        if device is not None:
            self.to_gpu(device)

    def to_gpu(self, device: int = 0):
        check(lib().pk_model_to_gpu(self._h, device))
        return self

    def set_decode_loop(self, mode):
        """pk_model_set_decode_loop: "phases" (default) | "persistent" | "graph" -- same results, different launch structure."""
        check(lib().pk_model_set_decode_loop(self._h, {"phases": 0, "persistent": 1, "graph": 2}[mode]))

    def close(self):
        if getattr(self, "_h", None):
            lib().pk_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # stage entry points ---------------------------------------------------------------------------------
    def mel(self, pcm, return_logmel=False):
        pcm = _c(pcm)
        if pcm.ndim == 1:
            pcm = pcm[None]
        B, n = pcm.shape
        nf = lib().pk_mel_num_frames(n)
        feats = np.empty((B, nf, self.cfg.mel_bins), np.float32)
        lm = np.empty((B, self.cfg.mel_bins, nf), np.float32) if return_logmel else None
        check(lib().pk_mel(self._h, _f(pcm), B, n, _f(feats), _f(lm) if return_logmel else None))
        return (feats, lm) if return_logmel else feats

    # ragged (mixed-length) forms: lists of per-clip arrays in, lists of per-clip arrays out (packed along time inside) -----------------
    def mel_ragged(self, clips, return_logmel=False):
        clips = [_c(c).ravel() for c in clips]
        off = np.zeros(len(clips) + 1, np.int64)
        off[1:] = np.cumsum([len(c) for c in clips])
        pcm = np.concatenate(clips)
        nf = [lib().pk_mel_num_frames(len(c)) for c in clips]
        F = self.cfg.mel_bins
        feats = np.empty((sum(nf), F), np.float32)
        lm = np.empty(sum(nf) * F, np.float32) if return_logmel else None
        check(lib().pk_mel_ragged(self._h, _f(pcm), off.ctypes.data_as(i64p), len(clips), _f(feats), _f(lm) if return_logmel else None))
        o = np.concatenate([[0], np.cumsum(nf)])
        fl = [feats[o[i]:o[i + 1]] for i in range(len(clips))]
        if not return_logmel:
            return fl
        return fl, [lm[o[i] * F:o[i + 1] * F].reshape(F, nf[i]) for i in range(len(clips))]

    def encode_ragged(self, feats_list, stop_layer=-1, stop_stage=0):
        tm = np.asarray([f.shape[0] for f in feats_list], np.int32)
        T = [lib().pk_encoder_num_frames(int(t)) for t in tm]
        feats = _c(np.concatenate([_c(f) for f in feats_list], axis=0))
        out = np.empty((sum(T), self.cfg.hidden_size), np.float32)
        check(lib().pk_encode_ragged(self._h, _f(feats), _i(tm), len(tm), stop_layer, stop_stage, _f(out)))
        o = np.concatenate([[0], np.cumsum(T)])
        return [out[o[i]:o[i + 1]] for i in range(len(tm))]

    def conformer_blocks_ragged(self, x_list, first_layer=0, n_layers=None):
        T = np.asarray([x.shape[0] for x in x_list], np.int32)
        x = _c(np.concatenate([_c(v) for v in x_list], axis=0))
        out = np.empty_like(x)
        n = self.cfg.num_layers - first_layer if n_layers is None else n_layers
        check(lib().pk_conformer_blocks_ragged(self._h, _f(x), _i(T), len(T), first_layer, n, _f(out)))
        o = np.concatenate([[0], np.cumsum(T)])
        return [out[o[i]:o[i + 1]] for i in range(len(T))]

    def ctc_decode_ragged(self, enc_list, return_logp=False):
        T = np.asarray([e.shape[0] for e in enc_list], np.int32)
        enc = _c(np.concatenate([_c(e) for e in enc_list], axis=0))
        B, tm = len(T), int(T.max())
        ids = np.zeros((B, tm), np.int32); st = np.zeros((B, tm), np.int32); en = np.zeros((B, tm), np.int32)
        cf = np.zeros((B, tm), np.float32); lens = np.zeros(B, np.int32)
        lp = np.empty((int(T.sum()), self.cfg.ctc_vocab_size), np.float32) if return_logp else None
        check(lib().pk_ctc_decode_ragged(self._h, _f(enc), _i(T), B, _i(ids), _i(lens), _i(st), _i(en), _f(cf), _f(lp) if return_logp else None))
        r = dict(ids=ids, lens=lens, start=st, end=en, conf=cf)
        if return_logp:
            o = np.concatenate([[0], np.cumsum(T)])
            r["logp"] = [lp[o[i]:o[i + 1]] for i in range(B)]
        return r

    def tdt_decode_ragged(self, enc_list, max_tokens=None):
        T = np.asarray([e.shape[0] for e in enc_list], np.int32)
        enc = _c(np.concatenate([_c(e) for e in enc_list], axis=0))
        B = len(T)
        mt = max_tokens or int(T.max()) * self.cfg.max_symbols_per_step
        ids = np.zeros((B, mt), np.int32); st = np.zeros((B, mt), np.int32); en = np.zeros((B, mt), np.int32)
        cf = np.zeros((B, mt), np.float32); lens = np.zeros(B, np.int32); steps = np.zeros(B, np.int32)
        check(lib().pk_tdt_decode_ragged(self._h, _f(enc), _i(T), B, mt, _i(ids), _i(lens), _i(st), _i(en), _f(cf), _i(steps)))
        r = dict(ids=ids, lens=lens, start=st, end=en, conf=cf, steps=steps)
        if not getattr(self, "_boosted", False):
            mg = np.zeros(B, np.float32)
            if lib().pk_decode_margins(self._h, _f(mg), B) == 0:
                r["min_margin"] = mg
        return r

    def subsample(self, feats):
        feats = _c(feats)
        B, Tm, _ = feats.shape
        out = np.empty((B, lib().pk_encoder_num_frames(Tm), self.cfg.hidden_size), np.float32)
        check(lib().pk_subsample(self._h, _f(feats), B, Tm, _f(out)))
        return out

    def encode(self, feats, stop_layer=-1, stop_stage=0):
        feats = _c(feats)
        B, Tm, _ = feats.shape
        out = np.empty((B, lib().pk_encoder_num_frames(Tm), self.cfg.hidden_size), np.float32)
        check(lib().pk_encode(self._h, _f(feats), B, Tm, stop_layer, stop_stage, _f(out)))
        return out

    def conformer_blocks(self, x, first_layer=0, n_layers=None):
        x = _c(x)
        B, T, _ = x.shape
        out = np.empty_like(x)
        n = self.cfg.num_layers - first_layer if n_layers is None else n_layers
        check(lib().pk_conformer_blocks(self._h, _f(x), B, T, first_layer, n, _f(out)))
        return out

    # phrase boosting (reference include/parakeet/phrase_boost.hpp) ------------------------------------------
    def set_boost_tokens(self, phrases, boost_score=5.0):
        """ContextTrie::insert of each token-id sequence; an empty list switches boosting off."""
        flat = np.asarray([t for p in phrases for t in p], np.int32)
        off = np.zeros(len(phrases) + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in phrases])
        check(lib().pk_set_boost_tokens(self._h, _i(flat) if len(flat) else _i(np.zeros(1, np.int32)), _i(off), len(phrases), boost_score))

    def set_boost_phrases(self, phrases, boost_score=5.0):
        """ContextTrie::build: Tokenizer::encode of each phrase (needs a vocabulary)."""
        arr = (C.c_char_p * max(1, len(phrases)))(*[p.encode() for p in phrases])
        check(lib().pk_set_boost_phrases(self._h, arr, len(phrases), boost_score))

    def boost_trie_size(self):
        return lib().pk_boost_trie_size(self._h)

    def tokenize(self, text):
        ids = np.zeros(4 * len(text.encode()) + 8, np.int32)
        n = lib().pk_tokenize(self._h, text.encode(), _i(ids), len(ids))
        return ids[:n].tolist()

    def transcribe_pcm(self, clips, decoder="tdt", timestamps=False, boost_phrases=(), boost_score=5.0):
        """pk_transcribe_pcm: Transcriber::transcribe (transcribe.hpp:91-180) on in-memory clips -> list of dicts."""
        return _transcribe(lib().pk_transcribe_pcm, self._h, clips, decoder, timestamps, boost_phrases, boost_score)

    def ctc_decode(self, enc, return_logp=False):
        enc = _c(enc)
        B, T, _ = enc.shape
        ids = np.zeros((B, T), np.int32); st = np.zeros((B, T), np.int32); en = np.zeros((B, T), np.int32)
        cf = np.zeros((B, T), np.float32); lens = np.zeros(B, np.int32)
        lp = np.empty((B, T, self.cfg.ctc_vocab_size), np.float32) if return_logp else None
        check(lib().pk_ctc_decode(self._h, _f(enc), B, T, _i(ids), _i(lens), _i(st), _i(en), _f(cf), _f(lp) if return_logp else None))
        r = dict(ids=ids, lens=lens, start=st, end=en, conf=cf)
        if return_logp:
            r["logp"] = lp
        return r

    def tdt_score(self, enc, labels, dur_idx):
        """pk_tdt_score: the TDT loop on ONE utterance enc[T][d] along the given decisions -> per-step label / duration log-probs."""
        enc = _c(enc)
        lab, dur = _c(labels, np.int32), _c(dur_idx, np.int32)
        n = len(lab)
        llp = np.zeros((n, self.cfg.vocab_size), np.float32); dlp = np.zeros((n, len(self.cfg.durations)), np.float32)
        done = C.c_int(0)
        check(lib().pk_tdt_score(self._h, _f(enc), enc.shape[0], _i(lab), _i(dur), n, _f(llp), _f(dlp), C.byref(done)))
        return dict(n=done.value, label_lp=llp[:done.value], dur_lp=dlp[:done.value])

    def tdt_decode(self, enc, max_tokens=None):
        enc = _c(enc)
        B, T, _ = enc.shape
        mt = max_tokens or T * self.cfg.max_symbols_per_step
        ids = np.zeros((B, mt), np.int32); st = np.zeros((B, mt), np.int32); en = np.zeros((B, mt), np.int32)
        cf = np.zeros((B, mt), np.float32); lens = np.zeros(B, np.int32); steps = np.zeros(B, np.int32)
        check(lib().pk_tdt_decode(self._h, _f(enc), B, T, mt, _i(ids), _i(lens), _i(st), _i(en), _f(cf), _i(steps)))
        r = dict(ids=ids, lens=lens, start=st, end=en, conf=cf, steps=steps)
        if not getattr(self, "_boosted", False):
            mg = np.zeros(B, np.float32)
            if lib().pk_decode_margins(self._h, _f(mg), B) == 0:
                r["min_margin"] = mg                         # smallest top-1 / top-2 label log-prob margin per utterance
        return r
