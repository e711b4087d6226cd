"""ctypes binding of the CPU oracle (oracle/libpk_oracle.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never
from the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False):
    """Compile the C restatement (and the real-reference text library when /root/reference exists)."""
    so = os.path.join(_HERE, "libpk_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pk_oracle.c", "pk_oracle.h", "pk_oracle_math.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "libpk_oracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libpk_ref_text.so")
    if os.path.exists("/root/reference/src/vocab.cpp") and (force or not os.path.exists(ref_so)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


class AudioConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("n_fft", C.c_int), ("win_length", C.c_int), ("hop_length", C.c_int),
                ("n_mels", C.c_int), ("f_min", C.c_float), ("f_max", C.c_float), ("normalize", C.c_int),
                ("window_centered", C.c_int), ("power_via_abs", C.c_int)]


def audio_config(n_mels=80, normalize=True, window_centered=False, power_via_abs=True):
    return AudioConfig(16000, 512, 400, 160, n_mels, 0.0, -1.0, int(normalize), int(window_centered), int(power_via_abs))


class OrcConfig(C.Structure):
    _fields_ = [("mel_bins", C.c_int), ("sub_channels", C.c_int), ("d_model", C.c_int), ("n_layers", C.c_int),
                ("n_heads", C.c_int), ("ffn", C.c_int), ("conv_k", C.c_int), ("vocab", C.c_int),
                ("pred_hidden", C.c_int), ("lstm_layers", C.c_int), ("joint_hidden", C.c_int),
                ("n_durations", C.c_int), ("durations", C.c_int * 8), ("blank_id", C.c_int),
                ("max_symbols", C.c_int), ("ln_eps", C.c_float), ("bn_eps", C.c_float),
                ("joint_pred_bias", C.c_int), ("gemm_bf16", C.c_int), ("joint_prefix", C.c_char * 32)]


f32p, i32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build()
    L = C.CDLL(so)
    f32p, i32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.orc_last_error.restype = C.c_char_p
    L.orc_get_max_threads.restype = C.c_int
    L.orc_sum64_f.restype = C.c_float
    L.orc_sum64_f.argtypes = [f32p, C.c_int64]
    L.orc_math_v.argtypes = [C.c_int, f32p, f32p, C.c_int64]
    L.orc_linear.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p]
    L.orc_linear_scalar.argtypes = L.orc_linear.argtypes
    L.orc_layer_norm.argtypes = [f32p, C.c_int64, C.c_int, f32p, f32p, C.c_float, f32p]
    L.orc_model_new.restype = C.c_void_p
    L.orc_model_new.argtypes = [C.POINTER(OrcConfig)]
    L.orc_model_free.argtypes = [C.c_void_p]
    L.orc_model_add.argtypes = [C.c_void_p, C.c_char_p, f32p, C.c_int, i64p]
    L.orc_mel_filterbank.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, f32p]
    L.orc_mel_num_frames.argtypes = [C.c_int64, C.c_int]
    L.orc_mel.argtypes = [C.POINTER(AudioConfig), f32p, C.c_int64, f32p, f32p]
    L.orc_pos_emb.argtypes = [C.c_int, C.c_int, f32p]
    L.orc_subsampled_len.argtypes = [C.c_int]
    L.orc_subsampling.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, f32p, f32p, f32p]
    L.orc_conformer_block.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, C.c_int, f32p, C.c_int]
    L.orc_encoder.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, f32p, f32p]
    L.orc_ctc_logprobs.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, f32p]
    L.orc_ctc_greedy.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f32p]
    L.orc_tdt_greedy.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p,
                                 f32p, i32p, f32p]
    L.orc_rnnt_greedy.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, f32p]
    _LIB = L
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _out(shape, dt=np.float32):
    """Output buffer, first-touched serially here (see xmalloc in pk_oracle.c)."""
    a = np.empty(shape, dt)
    a.fill(0)
    return a


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


MATH_FN = {"exp": 0, "log": 1, "tanh": 2, "sigmoid": 3, "silu": 4, "sqrt": 5, "rcp": 6}


def math_v(fn: str, x):
    x = _c(x)
    y = _out(x.shape)
    lib().orc_math_v(MATH_FN[fn], _f(x), _f(y), x.size)
    return y


def sum64(x):
    x = _c(x)
    return np.float32(lib().orc_sum64_f(_f(x), x.size))


def linear(A, W, bias=None, scalar=False):
    A, W = _c(A), _c(W)
    M, K = A.shape
    N = W.shape[0]
    out = _out((M, N), np.float32)
    b = _c(bias) if bias is not None else None
    fn = lib().orc_linear_scalar if scalar else lib().orc_linear
    fn(M, N, K, _f(A), _f(W), _f(b) if b is not None else None, _f(out))
    return out


def layer_norm(x, g, b, eps=1e-5):
    x, g, b = _c(x), _c(g), _c(b)
    y = _out(x.shape)
    lib().orc_layer_norm(_f(x), x.size // x.shape[-1], x.shape[-1], _f(g), _f(b), eps, _f(y))
    return y


def mel_filterbank(n_mels=80, n_freqs=257, sr=16000.0, f_min=0.0, f_max=8000.0):
    fb = _out((n_freqs, n_mels), np.float32)
    lib().orc_mel_filterbank(n_freqs, n_mels, sr, f_min, f_max, _f(fb))
    return fb


def mel(pcm, n_mels=80, return_logmel=False, **kw):
    """pcm[n] -> features [n_frames, n_mels]  (reference: preprocess_audio, src/audio.cpp:100-158)."""
    pcm = _c(pcm)
    ac = audio_config(n_mels=n_mels, **kw)
    nf = lib().orc_mel_num_frames(pcm.size, 160)
    out = _out((nf, n_mels), np.float32)
    tap = _out((n_mels, nf), np.float32) if return_logmel else None
    r = lib().orc_mel(C.byref(ac), _f(pcm), pcm.size, _f(out), _f(tap) if tap is not None else None)
    if r < 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return (out, tap) if return_logmel else out


def pos_emb(T, d):
    pe = _out((2 * T - 1, d), np.float32)
    lib().orc_pos_emb(T, d, _f(pe))
    return pe


def subsampled_len(n):
    return lib().orc_subsampled_len(n)


class Model:
    """Oracle model: reference tensor names -> numpy arrays (kept alive here)."""

    def __init__(self, cfg, weights: dict, joint_pred_bias=False, ln_eps=1e-5, bn_eps=1e-5, gemm_bf16=None):
        self.cfg = cfg
        oc = OrcConfig()
        oc.mel_bins, oc.sub_channels, oc.d_model = cfg.mel_bins, cfg.subsampling_channels, cfg.hidden_size
        oc.n_layers, oc.n_heads, oc.ffn, oc.conv_k = cfg.num_layers, cfg.num_heads, cfg.ffn_intermediate, cfg.conv_kernel_size
        oc.vocab, oc.pred_hidden, oc.lstm_layers, oc.joint_hidden = cfg.vocab_size, cfg.pred_hidden, cfg.num_lstm_layers, cfg.joint_hidden
        oc.n_durations = len(cfg.durations)
        for i, d in enumerate(cfg.durations):
            oc.durations[i] = d
        oc.blank_id, oc.max_symbols = cfg.blank_id, cfg.max_symbols_per_step
        oc.ln_eps, oc.bn_eps, oc.joint_pred_bias = ln_eps, bn_eps, int(joint_pred_bias)
        oc.gemm_bf16 = int(getattr(cfg, "gemm_bf16", False) if gemm_bf16 is None else gemm_bf16)
        oc.joint_prefix = cfg.joint_prefix.encode()
        self._h = lib().orc_model_new(C.byref(oc))
        ep = getattr(cfg, "encoder_prefix", "encoder_.")
        if ep != "encoder_." or getattr(cfg, "xscaling", False):
            lib().orc_model_set_encoder.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
            lib().orc_model_set_encoder(self._h, ep.encode(), int(getattr(cfg, "xscaling", False)))
        self._keep = {}
        for k, v in weights.items():
            a = _c(v)
            self._keep[k] = a
            shp = (C.c_int64 * max(1, a.ndim))(*a.shape)
            if lib().orc_model_add(self._h, k.encode(), _f(a), a.ndim, shp) != 0:
                raise RuntimeError(lib().orc_last_error().decode())

    def __del__(self):
        try:
            if self._h:
                lib().orc_model_free(self._h)
                self._h = None
        except Exception:
            pass

    def _chk(self, r):
        if r < 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return r

    def subsampling(self, feats, taps=False):
        feats = _c(feats)
        B, Tm, F = feats.shape
        T = subsampled_len(Tm)
        out = _out((B, T, self.cfg.hidden_size), np.float32)
        t1 = t3 = None
        if taps:
            H1, W1 = (Tm - 1) // 2 + 1, (F - 1) // 2 + 1
            H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
            H3, W3 = (H2 - 1) // 2 + 1, (W2 - 1) // 2 + 1
            t1 = _out((B, H1, W1, self.cfg.subsampling_channels), np.float32)
            t3 = _out((B, H3, W3, self.cfg.subsampling_channels), np.float32)
        self._chk(lib().orc_subsampling(self._h, _f(feats), B, Tm, _f(out), _f(t1) if taps else None, _f(t3) if taps else None))
        return (out, t1, t3) if taps else out

    def conformer_block(self, layer, x, pe=None, stop_after=0):
        x = _c(x).copy()
        B, T, d = x.shape
        pe = pos_emb(T, d) if pe is None else _c(pe)
        self._chk(lib().orc_conformer_block(self._h, layer, _f(x), B, T, _f(pe), stop_after))
        return x

    def encoder(self, feats, layer_taps=False):
        feats = _c(feats)
        B, Tm, _ = feats.shape
        T = subsampled_len(Tm)
        out = _out((B, T, self.cfg.hidden_size), np.float32)
        taps = _out((self.cfg.num_layers, B, T, self.cfg.hidden_size), np.float32) if layer_taps else None
        self._chk(lib().orc_encoder(self._h, _f(feats), B, Tm, _f(out), _f(taps) if layer_taps else None))
        return (out, taps) if layer_taps else out

    def transformer_encoder(self, x, prefix, n_layers, n_heads, pre_ln=True, has_final_norm=False, ln_eps=1e-5):
        """TransformerEncoder::forward (src/transformer.cpp:78-88) on x[B][T][d]; weights under `prefix` in this model."""
        x = _c(x).copy()
        B, T, d = x.shape
        lib().orc_transformer_encoder.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                                  C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int]
        self._chk(lib().orc_transformer_encoder(self._h, prefix.encode(), n_layers, n_heads, int(pre_ln), int(has_final_norm), ln_eps,
                                                _f(x), B, T, d))
        return x

    def sortformer_forward(self, feats, sf):
        """Sortformer::forward (src/sortformer.cpp:50-69): feats [B][Tm][mel] -> sigmoid speaker activities [B][T][S]."""
        feats = _c(feats)
        B, Tm, _ = feats.shape
        T = subsampled_len(Tm)
        probs = _out((B, T, sf.max_speakers), np.float32)
        L = lib()
        L.orc_sortformer_forward.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        self._chk(L.orc_sortformer_forward(self._h, _f(feats), B, Tm, sf.transformer_layers, sf.transformer_heads, int(sf.pre_ln),
                                           int(sf.has_final_norm), _f(probs)))
        return probs

    def ctc_logprobs(self, enc):
        enc = _c(enc)
        B, T, _ = enc.shape
        V = self.cfg.ctc_vocab_size
        lp = _out((B, T, V), np.float32)
        self._chk(lib().orc_ctc_logprobs(self._h, _f(enc), B, T, _f(lp)))
        return lp

    def tdt_greedy(self, enc, max_tokens=None, max_steps=0, first_logp=False, margin=False):
        enc = _c(enc)
        B, T, _ = enc.shape
        mt = max_tokens or (T * self.cfg.max_symbols_per_step)
        ids = np.zeros((B, mt), np.int32); st = np.zeros((B, mt), np.int32); en = np.zeros((B, mt), np.int32)
        cf = np.zeros((B, mt), np.float32); lens = np.zeros(B, np.int32); steps = np.zeros(B, np.int32)
        if margin:                              # per-utterance smallest top-1 / top-2 label log-prob margin
            L = lib()
            L.orc_tdt_greedy_margin.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f32p, i32p, f32p,
                                                f32p, i32p, C.c_int]
            mg = np.zeros(B, np.float32)
            cap = T * (self.cfg.max_symbols_per_step + 1) + 16
            smg = np.zeros((B, cap), np.float32); slab = np.full((B, cap), -1, np.int32)
            r = self._chk(L.orc_tdt_greedy_margin(self._h, _f(enc), B, T, mt, max_steps, _i(ids), _i(lens), _i(st), _i(en), _f(cf), _i(steps), _f(mg),
                                                  _f(smg), _i(slab), cap))
            return dict(ids=ids, lens=lens, start=st, end=en, conf=cf, steps=steps, overflow=bool(r), min_margin=mg, step_margin=smg,
                        step_label=slab)
        fl = np.zeros((B, self.cfg.vocab_size), np.float32) if first_logp else None
        r = self._chk(lib().orc_tdt_greedy(self._h, _f(enc), B, T, mt, max_steps, _i(ids), _i(lens), _i(st), _i(en),
                                           _f(cf), _i(steps), _f(fl) if first_logp else None))
        res = dict(ids=ids, lens=lens, start=st, end=en, conf=cf, steps=steps, overflow=bool(r))
        if first_logp:
            res["first_logp"] = fl
        return res

    def tdt_score(self, enc, labels=None, dur_idx=None, max_steps=None, rows=True):
        """orc_tdt_score: the TDT loop (src/tdt.cpp:62-106) on ONE utterance enc[T][d] along a GIVEN decision path (labels[k], dur_idx[k]) --
        or, labels=None, along its own greedy path -- returning every step's joint outputs: label log-probs [n][V], duration log-probs [n][D]
        and the decisions walked."""
        enc = _c(enc)
        T = enc.shape[0]
        V, D = self.cfg.vocab_size, len(self.cfg.durations)
        cap = int(max_steps or (len(labels) if labels is not None else T * (self.cfg.max_symbols_per_step + 1) + 16))
        L = lib()
        L.orc_tdt_score.argtypes = [C.c_void_p, f32p, C.c_int, i32p, i32p, C.c_int, i32p, i32p, f32p, f32p]
        lab_out = np.full(cap, -1, np.int32); dur_out = np.full(cap, -1, np.int32)
        llp = np.zeros((cap, V), np.float32) if rows else None
        dlp = np.zeros((cap, D), np.float32)
        li = _c(labels, np.int32) if labels is not None else None
        di = _c(dur_idx, np.int32) if labels is not None else None
        n = self._chk(L.orc_tdt_score(self._h, _f(enc), T, _i(li) if li is not None else None, _i(di) if di is not None else None, cap,
                                      _i(lab_out), _i(dur_out), _f(llp) if rows else None, _f(dlp)))
        return dict(n=n, labels=lab_out[:n], dur_idx=dur_out[:n], label_lp=(llp[:n] if rows else None), dur_lp=dlp[:n])

    def tdt_greedy_boosted(self, enc, trie, boost=5.0, max_tokens=None, max_steps=0):
        """tdt_greedy_decode(_with_timestamps)_boosted: src/phrase_boost.cpp:177-350."""
        enc = _c(enc)
        B, T, _ = enc.shape
        mt = max_tokens or (T * self.cfg.max_symbols_per_step)
        ids = np.zeros((B, mt), np.int32); st = np.zeros((B, mt), np.int32); en = np.zeros((B, mt), np.int32)
        cf = np.zeros((B, mt), np.float32); lens = np.zeros(B, np.int32); steps = np.zeros(B, np.int32)
        L = lib()
        L.orc_tdt_greedy_boosted.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, i32p, i32p,
                                             i32p, i32p, f32p, i32p]
        r = self._chk(L.orc_tdt_greedy_boosted(self._h, _f(enc), B, T, mt, max_steps, trie._h, boost, _i(ids), _i(lens), _i(st),
                                               _i(en), _f(cf), _i(steps)))
        return dict(ids=ids, lens=lens, start=st, end=en, conf=cf, steps=steps, overflow=bool(r))

    def rnnt_greedy(self, enc, max_tokens=None):
        enc = _c(enc)
        B, T, _ = enc.shape
        mt = max_tokens or (T * self.cfg.max_symbols_per_step)
        ids = np.zeros((B, mt), np.int32); st = np.zeros((B, mt), np.int32)
        cf = np.zeros((B, mt), np.float32); lens = np.zeros(B, np.int32)
        self._chk(lib().orc_rnnt_greedy(self._h, _f(enc), B, T, mt, _i(ids), _i(lens), _i(st), _f(cf)))
        return dict(ids=ids, lens=lens, start=st, conf=cf)


def ctc_greedy(logp, blank_id):
    """ctc_greedy_decode(+_with_timestamps): src/ctc.cpp:40-127."""
    logp = _c(logp)
    B, T, V = logp.shape
    ids = np.zeros((B, T), np.int32); st = np.zeros((B, T), np.int32); en = np.zeros((B, T), np.int32)
    cf = np.zeros((B, T), np.float32); lens = np.zeros(B, np.int32)
    lib().orc_ctc_greedy(_f(logp), B, T, V, blank_id, _i(ids), _i(lens), _i(st), _i(en), _f(cf))
    return dict(ids=ids, lens=lens, start=st, end=en, conf=cf)


def probs_to_segments(probs, threshold=0.5):
    """Sortformer::probs_to_segments (src/sortformer.cpp:71-113) on probs [T][S] -> list of (speaker, start_s, end_s)."""
    probs = _c(probs)
    T, S = probs.shape
    cap = S * (T // 2 + 2)
    spk = np.zeros(cap, np.int32); a = np.zeros(cap, np.float32); b = np.zeros(cap, np.float32)
    L = lib()
    L.orc_probs_to_segments.argtypes = [f32p, C.c_int, C.c_int, C.c_float, i32p, f32p, f32p]
    n = L.orc_probs_to_segments(_f(probs), T, S, threshold, _i(spk), _f(a), _f(b))
    return [(int(spk[i]), float(a[i]), float(b[i])) for i in range(n)]


class Trie:
    """ContextTrie (include/parakeet/phrase_boost.hpp:22-57, src/phrase_boost.cpp:9-66)."""

    def __init__(self, phrases=()):
        L = lib()
        L.orc_trie_new.restype = C.c_void_p
        L.orc_trie_free.argtypes = [C.c_void_p]
        L.orc_trie_insert.argtypes = [C.c_void_p, i32p, C.c_int]
        L.orc_trie_size.argtypes = [C.c_void_p]
        self._h = C.c_void_p(L.orc_trie_new())
        for p in phrases:
            self.insert(p)

    def insert(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        lib().orc_trie_insert(self._h, _i(ids) if len(ids) else None, len(ids))

    def size(self):
        return lib().orc_trie_size(self._h)

    def boosted_tokens(self, states, V=2048):
        """get_boosted_tokens (src/phrase_boost.cpp:39-50) -> sorted token ids."""
        st = np.ascontiguousarray(list(states), np.int32)
        flag = np.zeros(V, np.uint8)
        L = lib()
        L.orc_trie_boosted_tokens.argtypes = [C.c_void_p, i32p, C.c_int, C.c_void_p, C.c_int]
        L.orc_trie_boosted_tokens(self._h, _i(st), len(st), flag.ctypes.data_as(C.c_void_p), V)
        return set(np.nonzero(flag)[0].tolist())

    def advance(self, states, tok):
        """advance (src/phrase_boost.cpp:52-66) -> set of node ids (always holds the root)."""
        st = np.ascontiguousarray(list(states), np.int32)
        out = np.zeros(300, np.int32)
        L = lib()
        L.orc_trie_advance.argtypes = [C.c_void_p, i32p, C.c_int, C.c_int, i32p]
        n = L.orc_trie_advance(self._h, _i(st), len(st), tok, _i(out))
        return set(out[:n].tolist())

    def __del__(self):
        try:
            if self._h:
                lib().orc_trie_free(self._h)
                self._h = None
        except Exception:
            pass


def ctc_greedy_boosted(logp, blank_id, trie, boost=5.0):
    """ctc_greedy_decode(_with_timestamps)_boosted: src/phrase_boost.cpp:70-171."""
    logp = _c(logp)
    B, T, V = logp.shape
    ids = np.zeros((B, T), np.int32); st = np.zeros((B, T), np.int32); en = np.zeros((B, T), np.int32)
    cf = np.zeros((B, T), np.float32); lens = np.zeros(B, np.int32)
    L = lib()
    L.orc_ctc_greedy_boosted.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, i32p, i32p, i32p, i32p, f32p]
    L.orc_ctc_greedy_boosted(_f(logp), B, T, V, blank_id, trie._h, boost, _i(ids), _i(lens), _i(st), _i(en), _f(cf))
    return dict(ids=ids, lens=lens, start=st, end=en, conf=cf)


def set_threads(n):
    lib().orc_set_threads(n)


def max_threads():
    return lib().orc_get_max_threads()


class Stream:
    """One streaming session of the oracle: StreamingAudioPreprocessor + EncoderCache + StreamingDecodeState
    (reference src/audio.cpp:171-259, src/streaming_encoder.cpp:430-472, src/eou.cpp:17-98)."""

    def __init__(self, model: Model, att_context_left=70, att_context_right=0):
        L = lib()
        L.orc_stream_new.restype = C.c_void_p
        L.orc_stream_new.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_stream_free.argtypes = [C.c_void_p]
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_stream_mel.argtypes = [C.c_void_p, f32p, C.c_int, f32p]
        L.orc_stream_encode.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_int]
        L.orc_stream_decode.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, i32p, i32p, i32p, f32p]
        self.model = model
        self._h = L.orc_stream_new(model._h, att_context_left, att_context_right)

    def mel(self, pcm):
        pcm = _c(pcm)
        out = np.zeros((pcm.size // 160 + 8, self.model.cfg.mel_bins), np.float32)
        n = lib().orc_stream_mel(self._h, _f(pcm), pcm.size, _f(out))
        return out[:n]

    def encode(self, mel):
        mel = _c(mel)
        cap = mel.shape[0] // 8 + 2
        out = np.zeros((cap, self.model.cfg.hidden_size), np.float32)
        n = lib().orc_stream_encode(self._h, _f(mel), mel.shape[0], _f(out), cap)
        if n < 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out[:n]

    def decode(self, enc, margins=False):
        """margins=True: also step_label / step_margin -- the label chosen (blank included) and the top-1 / top-2 log-prob margin of every decision
        of the chunk, in order (what oracle/tolerance.py's first_divergence walks)."""
        enc = _c(enc)
        c = enc.shape[0]
        mt = max(1, c * self.model.cfg.max_symbols_per_step)
        ids = np.zeros(mt, np.int32); st = np.zeros(mt, np.int32); en = np.zeros(mt, np.int32); cf = np.zeros(mt, np.float32)
        if not margins:
            n = lib().orc_stream_decode(self._h, _f(enc), c, mt, _i(ids), _i(st), _i(en), _f(cf))
            if n < 0:
                raise RuntimeError(lib().orc_last_error().decode())
            return dict(ids=ids[:n], start=st[:n], end=en[:n], conf=cf[:n])
        L = lib()
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_stream_decode_ex.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, i32p, i32p, i32p, f32p, f32p, i32p, C.c_int, i32p]
        cap = c * (self.model.cfg.max_symbols_per_step + 1) + 16
        sm = np.zeros(cap, np.float32); sl = np.full(cap, -1, np.int32); ns = np.zeros(1, np.int32)
        n = L.orc_stream_decode_ex(self._h, _f(enc), c, mt, _i(ids), _i(st), _i(en), _f(cf), _f(sm), _i(sl), cap, _i(ns))
        if n < 0:
            raise RuntimeError(L.orc_last_error().decode())
        k = int(ns[0])
        return dict(ids=ids[:n], start=st[:n], end=en[:n], conf=cf[:n], step_label=sl[:k], step_margin=sm[:k])

    def score(self, enc, labels=None, dur_idx=None):
        """orc_stream_score: the chunk loop of rnnt_streaming_decode_chunk (src/eou.cpp:17-98) along a GIVEN decision path (labels[k], dur_idx[k];
        None: its own greedy path), the stream's LSTM state / last token carried in and out -> every step's joint outputs."""
        enc = _c(enc)
        c = enc.shape[0]
        cfg = self.model.cfg
        V, D = cfg.vocab_size, len(cfg.durations)
        cap = int(len(labels) if labels is not None else c * (cfg.max_symbols_per_step + 1) + 16)
        L = lib()
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_stream_score.argtypes = [C.c_void_p, f32p, C.c_int, i32p, i32p, C.c_int, i32p, i32p, f32p, f32p]
        lab_out = np.full(max(cap, 1), -1, np.int32); dur_out = np.full(max(cap, 1), -1, np.int32)
        llp = np.zeros((max(cap, 1), V), np.float32); dlp = np.zeros((max(cap, 1), D), np.float32)
        li = _c(labels, np.int32) if labels is not None else None
        di = _c(dur_idx, np.int32) if labels is not None else None
        n = L.orc_stream_score(self._h, _f(enc), c, _i(li) if li is not None else None, _i(di) if di is not None else None, cap,
                               _i(lab_out), _i(dur_out), _f(llp), _f(dlp))
        if n < 0:
            raise RuntimeError(L.orc_last_error().decode())
        return dict(n=n, labels=lab_out[:n], dur_idx=dur_out[:n], label_lp=llp[:n], dur_lp=dlp[:n])

    def sortformer_chunk(self, feats, sf):
        """Sortformer::diarize_chunk (src/sortformer.cpp:123-150) on feats [n_frames][mel] -> probs [c][S] (c may be 0)."""
        feats = _c(feats)
        cap = feats.shape[0] // 8 + 4
        probs = np.zeros((cap, sf.max_speakers), np.float32)
        L = lib()
        L.orc_sortformer_chunk.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        c = L.orc_sortformer_chunk(self._h, _f(feats), feats.shape[0], sf.transformer_layers, sf.transformer_heads, int(sf.pre_ln),
                                   int(sf.has_final_norm), _f(probs), cap)
        if c < 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return probs[:c]

    def push(self, pcm):
        """transcribe_chunk (src/nemotron.cpp:24-52): PCM chunk -> new tokens of this chunk (dict) or None."""
        m = self.mel(pcm)
        if m.shape[0] == 0:
            return None
        e = self.encode(m)
        if e.shape[0] == 0:
            return None
        return self.decode(e)

    def close(self):
        if self._h:
            lib().orc_stream_free(self._h)
            self._h = None
