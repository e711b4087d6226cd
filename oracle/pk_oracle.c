/*
 * oracle/pk_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the Parakeet hot path of Frikallo/parakeet.cpp
 * (mel front end -> FastConformer encoder -> CTC / TDT / RNNT greedy decode).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (parakeet.cpp_amd/csrc) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the floating-point arithmetic.  The
 * reference's arithmetic lives in the un-vendored `axiom` submodule
 * (/root/reference/.gitmodules:1-3, third_party/axiom is empty), so the
 * reference cannot be compiled or run here, and its tests hold no golden
 * numeric vectors for mel / encoder / joint.  What IS
 * pinned: the reference's CTC collapse KATs, position-embedding KATs, timestamp
 * and tokenizer KATs (tests/test_oracle_kats.py, against tests/test_all.cpp of
 * the reference), and a torch-CPU cross-check of every stage
 * (tests/test_oracle_vs_torch.py, modelled on scripts/compare_encoder.py and
 * scripts/compare_features.py of the reference).
 *
 * Every function cites the reference file:line it follows.  Where axiom's
 * behaviour is unknowable the choice is a switch in orc_config (A1..A6,
 * SURVEY.md section 8c) whose default is the PyTorch/NeMo semantics the reference
 * states it is matching (src/audio.cpp:14,116,134,141).
 *
 * Numerics contract: every dot product is a k-ordered fp32 fma
 * chain starting from 0 (bias added afterwards); every sum-reduction is the
 * canonical sum64 of pk_oracle_math.h; exp/log/tanh are the documented
 * polynomial evaluations; division and sqrt are IEEE.  Compiled with
 * -ffp-contract=off: all fusion is explicit.
 */
#include "pk_oracle.h"
#include "pk_oracle_math.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define ORC_AVX2 1
#endif
#ifdef _OPENMP
#include <omp.h>
#endif
#include <malloc.h>

/* Large scratch buffers are recycled from the heap instead of being mmap'd and
 * unmapped on every call, and are first-touched serially: in sandboxed hosts a
 * storm of parallel first-touch page faults is ~20x slower than the arithmetic. */
__attribute__((constructor)) static void orc_heap_init(void) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
}
static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n); abort(); }
    memset(p, 0, n);
    return p;
}

/* ------------------------------------------------------------------------- */
/* errors                                                                    */
/* ------------------------------------------------------------------------- */
static char g_err[512];
static int orc_fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
const char *orc_last_error(void) { return g_err; }

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* exported scalar math (for the math accuracy / device bit-parity tests)     */
/* ------------------------------------------------------------------------- */
void orc_math_v(int fn, const float *in, float *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float x = in[i], y;
        switch (fn) {
        case 0: y = orc_expf(x); break;
        case 1: y = orc_logf(x); break;
        case 2: y = orc_tanhf(x); break;
        case 3: y = orc_sigmoidf(x); break;
        case 4: y = orc_siluf(x); break;
        case 5: y = sqrtf(x); break;
        case 6: y = 1.0f / x; break;
        default: y = x;
        }
        out[i] = y;
    }
}
float orc_sum64_f(const float *x, int64_t n) { return orc_sum64(x, n, 1); }

/* ------------------------------------------------------------------------- */
/* GEMM core: out[m][n] = fma-chain over k = 0..K-1 of A[m][k]*WT[k][n]       */
/* (natural k order, accumulator starts at +0).  Vectorised across n, so      */
/* each output keeps its own sequential chain -> bit-identical to the scalar  */
/* loop and to an fp32 MFMA chain fed in the same k order.                    */
/* ------------------------------------------------------------------------- */
#define MR 6
#define NR 16

static void gemm_rows_scalar(int m0, int m1, int n0, int n1, int K, const float *A, int64_t lda,
                             const float *WT, int64_t ldw, float *out, int64_t ldo) {
    for (int m = m0; m < m1; ++m)
        for (int n = n0; n < n1; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[m * lda + k], WT[k * ldw + n], acc);
            out[m * ldo + n] = acc;
        }
}

#ifdef ORC_AVX2
static inline void mk_6x16(int K, const float *A, int64_t lda, int mr, const float *WT, int64_t ldw,
                           float *out, int64_t ldo) {
    const float *a[MR];
    for (int i = 0; i < MR; ++i) a[i] = A + (int64_t)(i < mr ? i : mr - 1) * lda;
    __m256 c00 = _mm256_setzero_ps(), c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00;
    __m256 c30 = c00, c31 = c00, c40 = c00, c41 = c00, c50 = c00, c51 = c00;
    for (int k = 0; k < K; ++k) {
        const __m256 w0 = _mm256_loadu_ps(WT + k * ldw);
        const __m256 w1 = _mm256_loadu_ps(WT + k * ldw + 8);
        __m256 x;
        x = _mm256_broadcast_ss(a[0] + k); c00 = _mm256_fmadd_ps(x, w0, c00); c01 = _mm256_fmadd_ps(x, w1, c01);
        x = _mm256_broadcast_ss(a[1] + k); c10 = _mm256_fmadd_ps(x, w0, c10); c11 = _mm256_fmadd_ps(x, w1, c11);
        x = _mm256_broadcast_ss(a[2] + k); c20 = _mm256_fmadd_ps(x, w0, c20); c21 = _mm256_fmadd_ps(x, w1, c21);
        x = _mm256_broadcast_ss(a[3] + k); c30 = _mm256_fmadd_ps(x, w0, c30); c31 = _mm256_fmadd_ps(x, w1, c31);
        x = _mm256_broadcast_ss(a[4] + k); c40 = _mm256_fmadd_ps(x, w0, c40); c41 = _mm256_fmadd_ps(x, w1, c41);
        x = _mm256_broadcast_ss(a[5] + k); c50 = _mm256_fmadd_ps(x, w0, c50); c51 = _mm256_fmadd_ps(x, w1, c51);
    }
    __m256 r0[MR] = {c00, c10, c20, c30, c40, c50};
    __m256 r1[MR] = {c01, c11, c21, c31, c41, c51};
    for (int i = 0; i < mr; ++i) {
        _mm256_storeu_ps(out + i * ldo, r0[i]);
        _mm256_storeu_ps(out + i * ldo + 8, r1[i]);
    }
}

/* M == 1 (decode GEMV): 64 columns per pass, 8 independent accumulators */
static inline void mk_1x64(int K, const float *a, const float *WT, int64_t ldw, float *out) {
    __m256 c0 = _mm256_setzero_ps(), c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int k = 0; k < K; ++k) {
        const __m256 x = _mm256_broadcast_ss(a + k);
        const float *w = WT + k * ldw;
        c0 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w), c0);
        c1 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w + 8), c1);
        c2 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w + 16), c2);
        c3 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w + 24), c3);
        c4 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w + 32), c4);
        c5 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w + 40), c5);
        c6 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w + 48), c6);
        c7 = _mm256_fmadd_ps(x, _mm256_loadu_ps(w + 56), c7);
    }
    _mm256_storeu_ps(out, c0); _mm256_storeu_ps(out + 8, c1); _mm256_storeu_ps(out + 16, c2); _mm256_storeu_ps(out + 24, c3);
    _mm256_storeu_ps(out + 32, c4); _mm256_storeu_ps(out + 40, c5); _mm256_storeu_ps(out + 48, c6); _mm256_storeu_ps(out + 56, c7);
}
#endif

/* Pack columns [n0, n0+16) of WT (K x N, row stride ldw) into a contiguous K x 16 panel. */
static void pack_panel(int K, const float *WT, int64_t ldw, int n0, float *panel) {
    for (int k = 0; k < K; ++k) memcpy(panel + (int64_t)k * NR, WT + (int64_t)k * ldw + n0, NR * sizeof(float));
}

static void gemm_core(int M, int N, int K, const float *A, int64_t lda, const float *WT, int64_t ldw,
                      float *out, int64_t ldo, int parallel) {
#ifdef ORC_AVX2
    if (M == 1) {
        int n0 = 0;
        for (; n0 + 64 <= N; n0 += 64) mk_1x64(K, A, WT + n0, ldw, out + n0);
        for (; n0 + NR <= N; n0 += NR) mk_6x16(K, A, lda, 1, WT + n0, ldw, out + n0, ldo);
        if (n0 < N) gemm_rows_scalar(0, 1, n0, N, K, A, lda, WT, ldw, out, ldo);
        return;
    }
    const int nfull = N / NR * NR;
    const int mblocks = (M + MR - 1) / MR;
    /* big problems: pack the weight panels once (contiguous K x 16), then sweep row chunks per panel */
    const int packed = (int64_t)M * K >= 4096 && ldw != NR;
    float *panels = NULL;
    if (packed) {
        panels = (float *)xmalloc((size_t)K * (nfull ? nfull : 1) * sizeof(float));
        for (int n0 = 0; n0 < nfull; n0 += NR) pack_panel(K, WT, ldw, n0, panels + (int64_t)n0 * K);
    }
    const int CH = 16; /* m-blocks per chunk: 96 rows of A stay cache-resident per panel */
    const int chunks = (mblocks + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 1) if (parallel && chunks >= 2 && !omp_in_parallel())
    for (int ch = 0; ch < chunks; ++ch) {
        const int mb0 = ch * CH, mb1 = (mb0 + CH) < mblocks ? (mb0 + CH) : mblocks;
        for (int n0 = 0; n0 < nfull; n0 += NR) {
            const float *pw = packed ? panels + (int64_t)n0 * K : WT + n0;
            const int64_t pld = packed ? NR : ldw;
            for (int mb = mb0; mb < mb1; ++mb) {
                const int m0 = mb * MR;
                const int mr = (M - m0) < MR ? (M - m0) : MR;
                mk_6x16(K, A + (int64_t)m0 * lda, lda, mr, pw, pld, out + (int64_t)m0 * ldo + n0, ldo);
            }
        }
        if (nfull < N) {
            const int m0 = mb0 * MR, m1 = (mb1 * MR) < M ? (mb1 * MR) : M;
            gemm_rows_scalar(m0, m1, nfull, N, K, A, lda, WT, ldw, out, ldo);
        }
    }
    free(panels);
#else
    (void)parallel;
    gemm_rows_scalar(0, M, 0, N, K, A, lda, WT, ldw, out, ldo);
#endif
}

/* Exported for tests: out = A[M,K] * W[N,K]^T (+bias), natural-k fma chains. */
void orc_linear(int M, int N, int K, const float *A, const float *W, const float *bias, float *out) {
    float *WT = (float *)xmalloc((size_t)K * N * sizeof(float));
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) WT[(int64_t)k * N + n] = W[(int64_t)n * K + k];
    gemm_core(M, N, K, A, K, WT, N, out, N, 1);
    if (bias)
        for (int64_t m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) out[m * N + n] = out[m * N + n] + bias[n];
    free(WT);
}
/* scalar reference of the same thing (validates the vector micro-kernel) */
void orc_linear_scalar(int M, int N, int K, const float *A, const float *W, const float *bias, float *out) {
    for (int64_t m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[m * K + k], W[(int64_t)n * K + k], acc);
            out[m * N + n] = bias ? acc + bias[n] : acc;
        }
}

/* ------------------------------------------------------------------------- */
/* model = config + named tensors (names are the reference's safetensors      */
/* names: scripts/convert_nemo.py:98-310)               */
/* ------------------------------------------------------------------------- */
typedef struct {
    char name[192];
    const float *data;
    int ndim;
    int64_t shape[4];
    float *wt; /* lazily built [K][N] transpose of a [N][K(,1)] matrix */
    float *wt16; /* the same with every element rounded to bf16 (gemm_bf16 mode) */
} orc_tensor;

struct orc_model {
    orc_config cfg;
    orc_tensor *t;
    int nt, cap;
    char enc_prefix[32];   /* module name of the FastConformer in the state dict: "encoder_." (default) or "nest_encoder_." (Sortformer) */
    int xscaling;          /* StreamingEncoderConfig::xscaling (streaming_encoder.cpp:402-406): x *= sqrt(d_model) after subsampling */
};
void orc_model_set_encoder(orc_model *m, const char *prefix, int xscaling) {
    snprintf(m->enc_prefix, sizeof m->enc_prefix, "%s", prefix ? prefix : "");
    m->xscaling = xscaling;
}

orc_model *orc_model_new(const orc_config *cfg) {
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    m->cfg = *cfg;
    return m;
}
void orc_model_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->nt; ++i) { free(m->t[i].wt); free(m->t[i].wt16); }
    free(m->t);
    free(m);
}
int orc_model_add(orc_model *m, const char *name, const float *data, int ndim, const int64_t *shape) {
    if (ndim > 4) return orc_fail("tensor %s: ndim %d > 4", name, ndim);
    if (m->nt == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 256;
        m->t = (orc_tensor *)realloc(m->t, (size_t)m->cap * sizeof(orc_tensor));
    }
    orc_tensor *t = &m->t[m->nt++];
    memset(t, 0, sizeof(*t));
    snprintf(t->name, sizeof(t->name), "%s", name);
    t->data = data;
    t->ndim = ndim;
    for (int i = 0; i < ndim; ++i) t->shape[i] = shape[i];
    return 0;
}
static orc_tensor *find(const orc_model *m, const char *name) {
    for (int i = 0; i < m->nt; ++i)
        if (strcmp(m->t[i].name, name) == 0) return &m->t[i];
    return NULL;
}
static orc_tensor *getf(const orc_model *m, const char *fmt, ...) {
    char name[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof(name), fmt, ap);
    va_end(ap);
    if (m->enc_prefix[0] && strncmp(name, "encoder_.", 9) == 0) {
        char alt[320];
        snprintf(alt, sizeof alt, "%s%s", m->enc_prefix, name + 9);
        memcpy(name, alt, sizeof name - 1);
        name[sizeof name - 1] = 0;
    }
    orc_tensor *t = find(m, name);
    if (!t) orc_fail("missing tensor '%s'", name);
    return t;
}
static int64_t numel(const orc_tensor *t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}
/* [N][K...] -> cached [K][N] */
static const float *wt_of(orc_tensor *t) {
    if (t->wt) return t->wt;
    const int64_t N = t->shape[0], K = numel(t) / N;
    float *w = (float *)xmalloc((size_t)(N * K) * sizeof(float));
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) w[k * N + n] = t->data[n * K + k];
    t->wt = w;
    return w;
}

/* fp32 -> nearest bf16 (ties to even), returned as the fp32 value it represents: the operand rounding of the product's
 * gemm_bf16 mode (weights rounded once at upload, activations by v_cvt_pk_bf16_f32 while they are staged). */
static inline float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}
static const float *wt16_of(orc_tensor *t) {
    if (t->wt16) return t->wt16;
    const float *w = wt_of(t);
    const int64_t n = numel(t);
    float *q = (float *)xmalloc((size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; ++i) q[i] = bf16_round(w[i]);
    t->wt16 = q;
    return q;
}

/* y[M][N] = x[M][K] * W^T + b   (Linear / 1x1 conv), W tensor [N][K(,1,1)].
 * bf16 != 0 (orc_config.gemm_bf16): both operands are rounded to bf16 first; products of two bf16 are exact in fp32 and
 * the accumulation stays a k-ordered fp32 chain (the device's MFMA sums in another order: compare with a tolerance). */
static int linear_t(int bf16, orc_tensor *W, const orc_tensor *b, int M, const float *x, int64_t ldx, float *y,
                    int64_t ldy, int parallel) {
    if (!W) return -1;
    const int N = (int)W->shape[0];
    const int K = (int)(numel(W) / N);
    if (bf16) {
        float *xq = (float *)xmalloc((size_t)M * K * sizeof(float));
        for (int64_t m = 0; m < M; ++m)
            for (int k = 0; k < K; ++k) xq[m * K + k] = bf16_round(x[m * ldx + k]);
        gemm_core(M, N, K, xq, K, wt16_of(W), N, y, ldy, parallel);
        free(xq);
    } else {
        gemm_core(M, N, K, x, ldx, wt_of(W), N, y, ldy, parallel);
    }
    if (b) {
        const float *bb = b->data;
        for (int64_t m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) y[m * ldy + n] = y[m * ldy + n] + bb[n];
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a1/a2: mel front end -- src/audio.cpp:100-158 (+ filterbank :40-94)         */
/* ------------------------------------------------------------------------- */
/* src/audio.cpp:24-36 */
static double hz_to_mel_slaney(double f) {
    return f < 1000.0 ? f / (200.0 / 3.0) : 15.0 + log(f / 1000.0) / 0.06875177742094912;
}
static double mel_to_hz_slaney(double m) {
    return m < 15.0 ? m * (200.0 / 3.0) : 1000.0 * exp((m - 15.0) * 0.06875177742094912);
}
/* src/audio.cpp:40-94: fb[f*n_mels+m], fp64 build, fp32 store */
void orc_mel_filterbank(int n_freqs, int n_mels, float sample_rate, float f_min, float f_max, float *fb) {
    const double mel_min = hz_to_mel_slaney(f_min), mel_max = hz_to_mel_slaney(f_max);
    double *hz = (double *)xmalloc((size_t)(n_mels + 2) * sizeof(double));
    for (int i = 0; i < n_mels + 2; ++i)
        hz[i] = mel_to_hz_slaney(mel_min + (double)i * (mel_max - mel_min) / (double)(n_mels + 1));
    for (int m = 0; m < n_mels; ++m) {
        const double left = hz[m], center = hz[m + 1], right = hz[m + 2];
        const double enorm = 2.0 / (right - left);
        for (int f = 0; f < n_freqs; ++f) {
            const double freq = (double)f * (double)sample_rate / (2.0 * (double)(n_freqs - 1));
            double val = 0.0;
            if (freq >= left && freq <= center && center > left)
                val = (freq - left) / (center - left);
            else if (freq > center && freq <= right && right > center)
                val = (right - freq) / (right - center);
            fb[f * n_mels + m] = (float)(val * enorm);
        }
    }
    free(hz);
}

int orc_mel_num_frames(int64_t n_samples, int hop) { return (int)(1 + n_samples / hop); }

/* FFT-512 specification (shared in writing with the HIP kernel):
 * radix-2 decimation-in-time, bit-reversed input order, stage s has half-size
 * h = 2^(s-1); butterfly (a,b=a+h) with twiddle w = tw[j*(N/(2h))]:
 *   tr = fma(-wi, bi, wr*br);  ti = fma(wi, br, wr*bi);
 *   b' = a - t;  a' = a + t.
 * tw[k] = ( (float)cos(2*pi*k/N), (float)(-sin(2*pi*k/N)) ), double-evaluated. */
static void fft_twiddles(int n, float *wr, float *wi) {
    for (int k = 0; k < n / 2; ++k) {
        const double a = 2.0 * M_PI * (double)k / (double)n;
        wr[k] = (float)cos(a);
        wi[k] = (float)(-sin(a));
    }
}
static void fft_radix2(int n, int logn, float *re, float *im, const float *wr, const float *wi) {
    for (int i = 0; i < n; ++i) {
        int j = 0;
        for (int b = 0; b < logn; ++b) j |= ((i >> b) & 1) << (logn - 1 - b);
        if (j > i) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int h = 1; h < n; h <<= 1) {
        const int step = n / (2 * h);
        for (int base = 0; base < n; base += 2 * h)
            for (int j = 0; j < h; ++j) {
                const int a = base + j, b = a + h;
                const float cr = wr[j * step], ci = wi[j * step];
                const float tr = fmaf(-ci, im[b], cr * re[b]);
                const float ti = fmaf(ci, re[b], cr * im[b]);
                const float ar = re[a], ai = im[a];
                re[b] = ar - tr; im[b] = ai - ti;
                re[a] = ar + tr; im[a] = ai + ti;
            }
    }
}

/*
 * src/audio.cpp:100-158.  pcm[n] -> out[n_frames][n_mels] (the reference's
 * (1, n_frames, n_mels) tensor).  Optional taps: logmel[n_mels][n_frames].
 */
int orc_mel(const orc_audio_config *ac, const float *pcm, int64_t n, float *out, float *logmel_tap) {
    const int n_fft = ac->n_fft, hop = ac->hop_length, win = ac->win_length, n_mels = ac->n_mels;
    if (n < 2) return orc_fail("orc_mel: need >= 2 samples");
    if (n_fft != 512) return orc_fail("orc_mel: n_fft must be 512");
    const int n_freqs = n_fft / 2 + 1;
    const int pad = n_fft / 2;
    if (n <= pad) return orc_fail("orc_mel: reflect padding needs n > n_fft/2");
    const int n_frames = orc_mel_num_frames(n, hop);

    /* 1. preemphasis  src/audio.cpp:104-114 */
    float *pre = (float *)xmalloc((size_t)n * sizeof(float));
    pre[0] = pcm[0];
    for (int64_t i = 1; i < n; ++i) {
        const float t = 0.97f * pcm[i - 1];
        pre[i] = pcm[i] - t;
    }
    /* 2. window: symmetric Hann(win) zero-padded to n_fft  src/audio.cpp:117 ; placement = switch A1 */
    float *w = (float *)calloc((size_t)n_fft, sizeof(float));
    const int off = ac->window_centered ? (n_fft - win) / 2 : 0;
    for (int k = 0; k < win; ++k) w[off + k] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)k / (double)(win - 1)));
    float *twr = (float *)xmalloc((size_t)n_fft / 2 * sizeof(float));
    float *twi = (float *)xmalloc((size_t)n_fft / 2 * sizeof(float));
    fft_twiddles(n_fft, twr, twi);
    /* 4a. filterbank  src/audio.cpp:127-130 */
    const float f_max = ac->f_max > 0 ? ac->f_max : (float)ac->sample_rate / 2.0f;
    float *fb = (float *)xmalloc((size_t)n_freqs * n_mels * sizeof(float));
    orc_mel_filterbank(n_freqs, n_mels, (float)ac->sample_rate, ac->f_min, f_max, fb);

    float *logmel = (float *)xmalloc((size_t)n_mels * n_frames * sizeof(float));
#pragma omp parallel
    {
        float *re = (float *)xmalloc((size_t)n_fft * sizeof(float));
        float *im = (float *)xmalloc((size_t)n_fft * sizeof(float));
        float *pw = (float *)xmalloc((size_t)n_freqs * sizeof(float));
#pragma omp for schedule(static)
        for (int t = 0; t < n_frames; ++t) {
            /* center=true, reflect padding  src/audio.cpp:119-120 */
            for (int k = 0; k < n_fft; ++k) {
                int64_t idx = (int64_t)t * hop + k - pad;
                if (idx < 0) idx = -idx;
                if (idx >= n) idx = 2 * (n - 1) - idx;
                re[k] = pre[idx] * w[k];
                im[k] = 0.0f;
            }
            fft_radix2(n_fft, 9, re, im, twr, twi);
            /* 3. power = abs()^2  src/audio.cpp:123-124 ; switch A2 */
            for (int f = 0; f < n_freqs; ++f) {
                const float s = fmaf(re[f], re[f], im[f] * im[f]);
                if (ac->power_via_abs) {
                    const float mag = sqrtf(s);
                    pw[f] = mag * mag;
                } else {
                    pw[f] = s;
                }
            }
            /* 4b. mel_fb^T @ power, 5. log(x + 2^-24)  src/audio.cpp:132-136 */
            for (int m = 0; m < n_mels; ++m) {
                float acc = 0.0f;
                for (int f = 0; f < n_freqs; ++f) acc = fmaf(fb[f * n_mels + m], pw[f], acc);
                logmel[(int64_t)m * n_frames + t] = orc_logf(acc + 5.96046448e-8f);
            }
        }
        free(re); free(im); free(pw);
    }
    if (logmel_tap) memcpy(logmel_tap, logmel, (size_t)n_mels * n_frames * sizeof(float));
    /* 6. per-bin normalisation (unbiased variance)  src/audio.cpp:139-149 ; 7. transpose :156 */
    float *cen = (float *)xmalloc((size_t)n_frames * sizeof(float));
    float *sq = (float *)xmalloc((size_t)n_frames * sizeof(float));
    for (int m = 0; m < n_mels; ++m) {
        const float *row = logmel + (int64_t)m * n_frames;
        if (ac->normalize) {
            const float mean = orc_sum64(row, n_frames, 1) / (float)n_frames;
            for (int t = 0; t < n_frames; ++t) { cen[t] = row[t] - mean; sq[t] = cen[t] * cen[t]; }
            const float var = orc_sum64(sq, n_frames, 1) / (float)(n_frames - 1);
            const float den = sqrtf(var) + 1e-5f;
            for (int t = 0; t < n_frames; ++t) out[(int64_t)t * n_mels + m] = cen[t] / den;
        } else {
            for (int t = 0; t < n_frames; ++t) out[(int64_t)t * n_mels + m] = row[t];
        }
    }
    free(cen); free(sq); free(logmel); free(fb); free(twr); free(twi); free(w); free(pre);
    return n_frames;
}

/* ------------------------------------------------------------------------- */
/* a4: sinusoidal_position_embedding -- src/encoder.cpp:9-30 (float math)     */
/* ------------------------------------------------------------------------- */
void orc_pos_emb(int seq_len, int d_model, float *pe) {
    const int total = 2 * seq_len - 1;
    for (int p = 0; p < total; ++p) {
        const float position = (float)(seq_len - 1 - p);
        for (int i = 0; i < d_model; i += 2) {
            const float div_term = expf((float)i * (-logf(10000.0f) / (float)d_model));
            pe[(int64_t)p * d_model + i] = sinf(position * div_term);
            if (i + 1 < d_model) pe[(int64_t)p * d_model + i + 1] = cosf(position * div_term);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a3: ConvSubsampling::forward -- src/encoder.cpp:219-241                    */
/* ------------------------------------------------------------------------- */
static int sub_len(int n) { return (n - 1) / 2 + 1; } /* floor((n+2-3)/2)+1 */
int orc_subsampled_len(int n_mel_frames) { return sub_len(sub_len(sub_len(n_mel_frames))); }

/* depthwise 3x3 stride-2 pad-1 on channels-last [H][W][C]; taps in (ky,kx) order; bias after */
static void dw3x3s2(const float *in, int H, int W, int C, const float *wt /*[C][1][3][3]*/, const float *bias,
                    float *out, int Ho, int Wo) {
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int c = 0; c < C; ++c) {
                float acc = 0.0f;
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = 2 * y + ky - 1;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = 2 * x + kx - 1;
                        if (ix < 0 || ix >= W) continue;
                        acc = fmaf(wt[c * 9 + ky * 3 + kx], in[((int64_t)iy * W + ix) * C + c], acc);
                    }
                }
                out[((int64_t)y * Wo + x) * C + c] = acc + bias[c];
            }
}

/* feats[B][Tm][F] -> out[B][T][d].  taps (optional): after conv1 / after conv3+ReLU, channels-last.
 * Utterances are independent: the batch loop is the (only) parallel loop. */
int orc_subsampling(orc_model *m, const float *feats, int B, int Tm, float *out, float *tap_conv1,
                    float *tap_stage3) {
    const orc_config *c = &m->cfg;
    const int F = c->mel_bins, C = c->sub_channels, d = c->d_model;
    orc_tensor *c1w = getf(m, "encoder_.subsampling_.conv1_.weight"), *c1b = getf(m, "encoder_.subsampling_.conv1_.bias");
    orc_tensor *d1w = getf(m, "encoder_.subsampling_.dw1_.weight"), *d1b = getf(m, "encoder_.subsampling_.dw1_.bias");
    orc_tensor *c2w = getf(m, "encoder_.subsampling_.conv2_.weight"), *c2b = getf(m, "encoder_.subsampling_.conv2_.bias");
    orc_tensor *d2w = getf(m, "encoder_.subsampling_.dw2_.weight"), *d2b = getf(m, "encoder_.subsampling_.dw2_.bias");
    orc_tensor *c3w = getf(m, "encoder_.subsampling_.conv3_.weight"), *c3b = getf(m, "encoder_.subsampling_.conv3_.bias");
    orc_tensor *pw = getf(m, "encoder_.subsampling_.proj_.weight"), *pb = getf(m, "encoder_.subsampling_.proj_.bias");
    if (!c1w || !c1b || !d1w || !d1b || !c2w || !c2b || !d2w || !d2b || !c3w || !c3b || !pw || !pb) return -1;
    const int H1 = sub_len(Tm), W1 = sub_len(F), H2 = sub_len(H1), W2 = sub_len(W1), H3 = sub_len(H2), W3 = sub_len(W2);
    if ((int64_t)C * W3 != pw->shape[1]) return orc_fail("subsampling proj expects %lld inputs, got %d", (long long)pw->shape[1], C * W3);
    wt_of(c2w); wt_of(c3w); wt_of(pw);
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        float *a1 = (float *)xmalloc((size_t)H1 * W1 * C * sizeof(float));
        float *a2 = (float *)xmalloc((size_t)H2 * W2 * C * sizeof(float));
        float *a3 = (float *)xmalloc((size_t)H2 * W2 * C * sizeof(float));
        float *a4 = (float *)xmalloc((size_t)H3 * W3 * C * sizeof(float));
        float *a5 = (float *)xmalloc((size_t)H3 * W3 * C * sizeof(float));
        float *flat = (float *)xmalloc((size_t)H3 * W3 * C * sizeof(float));
        const float *x = feats + (int64_t)b * Tm * F;
        /* conv1 (1->C, 3x3, s2, p1) + ReLU  src/encoder.cpp:223-224 */
        for (int y = 0; y < H1; ++y)
            for (int xx = 0; xx < W1; ++xx)
                for (int ch = 0; ch < C; ++ch) {
                    float acc = 0.0f;
                    for (int ky = 0; ky < 3; ++ky) {
                        const int iy = 2 * y + ky - 1;
                        if (iy < 0 || iy >= Tm) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int ix = 2 * xx + kx - 1;
                            if (ix < 0 || ix >= F) continue;
                            acc = fmaf(c1w->data[ch * 9 + ky * 3 + kx], x[(int64_t)iy * F + ix], acc);
                        }
                    }
                    const float v = acc + c1b->data[ch];
                    a1[((int64_t)y * W1 + xx) * C + ch] = v > 0.0f ? v : 0.0f;
                }
        if (tap_conv1) memcpy(tap_conv1 + (int64_t)b * H1 * W1 * C, a1, (size_t)H1 * W1 * C * sizeof(float));
        /* dw1 -> conv2 (1x1) -> ReLU  src/encoder.cpp:226-228 */
        dw3x3s2(a1, H1, W1, C, d1w->data, d1b->data, a2, H2, W2);
        linear_t(m->cfg.gemm_bf16, c2w, c2b, H2 * W2, a2, C, a3, C, 0);
        for (int64_t i = 0; i < (int64_t)H2 * W2 * C; ++i) a3[i] = a3[i] > 0.0f ? a3[i] : 0.0f;
        /* dw2 -> conv3 (1x1) -> ReLU  src/encoder.cpp:230-232 */
        dw3x3s2(a3, H2, W2, C, d2w->data, d2b->data, a4, H3, W3);
        linear_t(m->cfg.gemm_bf16, c3w, c3b, H3 * W3, a4, C, a5, C, 0);
        for (int64_t i = 0; i < (int64_t)H3 * W3 * C; ++i) a5[i] = a5[i] > 0.0f ? a5[i] : 0.0f;
        if (tap_stage3) memcpy(tap_stage3 + (int64_t)b * H3 * W3 * C, a5, (size_t)H3 * W3 * C * sizeof(float));
        /* permute(0,2,1,3)+reshape: feature index = c*W3 + f  src/encoder.cpp:235-238 */
        for (int t = 0; t < H3; ++t)
            for (int f = 0; f < W3; ++f)
                for (int ch = 0; ch < C; ++ch) flat[(int64_t)t * C * W3 + ch * W3 + f] = a5[((int64_t)t * W3 + f) * C + ch];
        /* proj_  src/encoder.cpp:240 */
        linear_t(m->cfg.gemm_bf16, pw, pb, H3, flat, (int64_t)C * W3, out + (int64_t)b * H3 * d, d, 0);
        free(a1); free(a2); free(a3); free(a4); free(a5); free(flat);
    }
    return H3;
}

/* ------------------------------------------------------------------------- */
/* LayerNorm (axiom nn::LayerNorm, default eps -- switch A3)                  */
/* ------------------------------------------------------------------------- */
static void layer_norm(const float *x, int64_t rows, int d, const float *g, const float *b, float eps, float *y) {
    float *tmp = (float *)malloc((size_t)d * sizeof(float));
    for (int64_t r = 0; r < rows; ++r) {
        const float *xr = x + r * d;
        float *yr = y + r * d;
        const float mean = orc_sum64(xr, d, 1) / (float)d;
        for (int i = 0; i < d; ++i) { const float c = xr[i] - mean; tmp[i] = c * c; }
        const float var = orc_sum64(tmp, d, 1) / (float)d;
        const float rstd = 1.0f / sqrtf(var + eps);
        for (int i = 0; i < d; ++i) yr[i] = fmaf((xr[i] - mean) * rstd, g[i], b[i]);
    }
    free(tmp);
}
void orc_layer_norm(const float *x, int64_t rows, int d, const float *g, const float *b, float eps, float *y) {
    layer_norm(x, rows, d, g, b, eps, y);
}

/* a5: FeedForward::forward -- src/encoder.cpp:39-46 : x + 0.5*fc2(silu(fc1(LN(x)))) */
static int feed_forward(orc_model *m, int layer, const char *which, float *x, int64_t rows) {
    const orc_config *c = &m->cfg;
    const int d = c->d_model, ffn = c->ffn;
    orc_tensor *ng = getf(m, "encoder_.layers_.%d.%s.norm_.weight", layer, which);
    orc_tensor *nb = getf(m, "encoder_.layers_.%d.%s.norm_.bias", layer, which);
    orc_tensor *w1 = getf(m, "encoder_.layers_.%d.%s.fc1_.weight", layer, which);
    orc_tensor *b1 = getf(m, "encoder_.layers_.%d.%s.fc1_.bias", layer, which);
    orc_tensor *w2 = getf(m, "encoder_.layers_.%d.%s.fc2_.weight", layer, which);
    orc_tensor *b2 = getf(m, "encoder_.layers_.%d.%s.fc2_.bias", layer, which);
    if (!ng || !nb || !w1 || !b1 || !w2 || !b2) return -1;
    float *n = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *h = (float *)xmalloc((size_t)rows * ffn * sizeof(float));
    float *y = (float *)xmalloc((size_t)rows * d * sizeof(float));
    layer_norm(x, rows, d, ng->data, nb->data, c->ln_eps, n);
    linear_t(m->cfg.gemm_bf16, w1, b1, (int)rows, n, d, h, ffn, 0);
    for (int64_t i = 0; i < rows * ffn; ++i) h[i] = orc_siluf(h[i]);
    linear_t(m->cfg.gemm_bf16, w2, b2, (int)rows, h, ffn, y, d, 0);
    for (int64_t i = 0; i < rows * d; ++i) x[i] = x[i] + y[i] * 0.5f;
    free(n); free(h); free(y);
    return 0;
}

/* The attention of the tolerance-class mode (gemm_bf16) for head sizes 64 / 128 -- the specification kernels/attention_bf16.hip implements:
 * q, k, v and the projected position table are rounded to bf16 (their GEMM epilogues store them so); the query tile is biased ONCE,
 * qu = bf16(q + u); content[i][j] = qu_i . k_j; position[i][p] = qu_i . P_p + c[p] with c[p] = (v - u) . P_p in fp32 (the identity
 * (q + v) . P = (q + u) . P + (v - u) . P keeps one biased copy); scores = (content + position) * scale; e_j = exp(score_j - max) in fp32;
 * ctx = (sum_j bf16(e_j) v_j) / (sum_j e_j) -- the probabilities enter the product rounded, the normaliser does not -- and ctx is stored as
 * bf16 (out_proj's operand).  Accumulation is fp32 throughout; its ORDER differs on the GPU (MFMA blocks, online softmax), which is why this
 * mode is compared within a tolerance.  Other head sizes keep the fp32 attention on the bf16-GEMM outputs. */
static int attn_bf16_spec(const orc_config *c) {
    const int hd = c->d_model / c->n_heads;
    return c->gemm_bf16 && (hd == 64 || hd == 128);
}

/* pos_proj_(pos_emb) (no bias), split by head and transposed: PT[h][k][p]  src/encoder.cpp:148-151.
 * Batch-independent (the reference recomputes it per call). */
static float *pos_proj_heads(orc_model *m, int layer, int T, const float *pos_emb, int round16 /* store the table as bf16 (attn_bf16_spec) */) {
    const orc_config *c = &m->cfg;
    const int d = c->d_model, H = c->n_heads, hd = d / H, P = 2 * T - 1;
    orc_tensor *wp = getf(m, "encoder_.layers_.%d.attn_.pos_proj_.weight", layer);
    if (!wp) return NULL;
    float *pp = (float *)xmalloc((size_t)P * d * sizeof(float));
    linear_t(m->cfg.gemm_bf16, wp, NULL, P, pos_emb, d, pp, d, 0);
    if (round16)                                                    /* the table is stored as bf16 (the pos_proj GEMM's epilogue rounds it) */
        for (int64_t i = 0; i < (int64_t)P * d; ++i) pp[i] = bf16_round(pp[i]);
    float *PT = (float *)xmalloc((size_t)H * hd * P * sizeof(float));
    for (int h = 0; h < H; ++h)
        for (int kk = 0; kk < hd; ++kk)
            for (int p = 0; p < P; ++p) PT[((int64_t)h * hd + kk) * P + p] = pp[(int64_t)p * d + h * hd + kk];
    free(pp);
    return PT;
}

/* a6: ConformerAttention::forward -- src/encoder.cpp:180-186,111-178 ; rel_shift :85-109 in closed form
 * pos[i][j] = (q_i + v) . P[j - i + T - 1]   (SURVEY.md Appendix A.4).  One utterance: x[T][d]. */
static int attention(orc_model *m, int layer, float *x, int T, const float *PT) {
    const orc_config *c = &m->cfg;
    const int d = c->d_model, H = c->n_heads, hd = d / H, P = 2 * T - 1;
    const int64_t rows = T;
    orc_tensor *ng = getf(m, "encoder_.layers_.%d.attn_.norm_.weight", layer), *nb = getf(m, "encoder_.layers_.%d.attn_.norm_.bias", layer);
    orc_tensor *wq = getf(m, "encoder_.layers_.%d.attn_.mha_.q_proj.weight", layer), *bq = getf(m, "encoder_.layers_.%d.attn_.mha_.q_proj.bias", layer);
    orc_tensor *wk = getf(m, "encoder_.layers_.%d.attn_.mha_.k_proj.weight", layer), *bk = getf(m, "encoder_.layers_.%d.attn_.mha_.k_proj.bias", layer);
    orc_tensor *wv = getf(m, "encoder_.layers_.%d.attn_.mha_.v_proj.weight", layer), *bv = getf(m, "encoder_.layers_.%d.attn_.mha_.v_proj.bias", layer);
    orc_tensor *wo = getf(m, "encoder_.layers_.%d.attn_.mha_.out_proj.weight", layer), *bo = getf(m, "encoder_.layers_.%d.attn_.mha_.out_proj.bias", layer);
    orc_tensor *pu = getf(m, "encoder_.layers_.%d.attn_.pos_bias_u_", layer), *pv = getf(m, "encoder_.layers_.%d.attn_.pos_bias_v_", layer);
    if (!ng || !nb || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !pu || !pv) return -1;
    float *n = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *q = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *k = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *v = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *ctx = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *y = (float *)xmalloc((size_t)rows * d * sizeof(float));
    layer_norm(x, rows, d, ng->data, nb->data, c->ln_eps, n);      /* :182 */
    linear_t(m->cfg.gemm_bf16, wq, bq, (int)rows, n, d, q, d, 0);                    /* :120-122 */
    linear_t(m->cfg.gemm_bf16, wk, bk, (int)rows, n, d, k, d, 0);
    linear_t(m->cfg.gemm_bf16, wv, bv, (int)rows, n, d, v, d, 0);
    const float scale = 1.0f / sqrtf((float)hd);                   /* :126 */
    const int spec16 = attn_bf16_spec(c);
    if (spec16)
        for (int64_t i = 0; i < rows * d; ++i) { q[i] = bf16_round(q[i]); k[i] = bf16_round(k[i]); v[i] = bf16_round(v[i]); }
    float *cvec = (float *)xmalloc((size_t)P * sizeof(float));
    float *qu = (float *)xmalloc((size_t)T * hd * sizeof(float));
    float *qv = (float *)xmalloc((size_t)T * hd * sizeof(float));
    float *KT = (float *)xmalloc((size_t)hd * T * sizeof(float));
    float *Vh = (float *)xmalloc((size_t)T * hd * sizeof(float));
    float *cs = (float *)xmalloc((size_t)T * T * sizeof(float));
    float *ps = (float *)xmalloc((size_t)T * P * sizeof(float));
    float *pr = (float *)xmalloc((size_t)T * T * sizeof(float));
    float *oh = (float *)xmalloc((size_t)T * hd * sizeof(float));
    for (int h = 0; h < H; ++h) {
        for (int i = 0; i < T; ++i)
            for (int kk = 0; kk < hd; ++kk) {
                const float qq = q[(int64_t)i * d + h * hd + kk];
                qu[i * hd + kk] = qq + pu->data[h * hd + kk];                    /* :141-145 */
                qv[i * hd + kk] = qq + pv->data[h * hd + kk];
                if (spec16) { qu[i * hd + kk] = bf16_round(qu[i * hd + kk]); qv[i * hd + kk] = qu[i * hd + kk]; }   /* one biased copy */
                KT[(int64_t)kk * T + i] = k[(int64_t)i * d + h * hd + kk];
                Vh[i * hd + kk] = v[(int64_t)i * d + h * hd + kk];
            }
        gemm_core(T, T, hd, qu, hd, KT, T, cs, T, 0);                             /* content :145 */
        gemm_core(T, P, hd, qv, hd, PT + (int64_t)h * hd * P, P, ps, P, 0);       /* pos :154 */
        if (spec16) {                                                             /* c[p] = (v - u) . P_p, natural k, explicit fma */
            const float *Ph = PT + (int64_t)h * hd * P;
            for (int p = 0; p < P; ++p) {
                float acc = 0.0f;
                for (int kk = 0; kk < hd; ++kk) acc = fmaf(pv->data[h * hd + kk] - pu->data[h * hd + kk], Ph[(int64_t)kk * P + p], acc);
                cvec[p] = acc;
            }
        }
        float *rsum = spec16 ? (float *)xmalloc((size_t)T * sizeof(float)) : NULL;
        for (int i = 0; i < T; ++i) {
            float *row = pr + (int64_t)i * T;
            float mx = -INFINITY;
            for (int j = 0; j < T; ++j) {
                const int p = j - i + T - 1;
                const float pos = spec16 ? ps[(int64_t)i * P + p] + cvec[p] : ps[(int64_t)i * P + p];
                const float s = (cs[(int64_t)i * T + j] + pos) * scale;            /* :157-160 */
                row[j] = s;
                mx = s > mx ? s : mx;
            }
            for (int j = 0; j < T; ++j) row[j] = orc_expf(row[j] - mx);            /* softmax :168 */
            const float sum = orc_sum64(row, T, 1);
            if (spec16) {
                rsum[i] = sum;
                for (int j = 0; j < T; ++j) row[j] = bf16_round(row[j]);
            } else {
                for (int j = 0; j < T; ++j) row[j] = row[j] / sum;
            }
        }
        gemm_core(T, hd, T, pr, T, Vh, hd, oh, hd, 0);                            /* :171 */
        for (int i = 0; i < T; ++i)
            for (int kk = 0; kk < hd; ++kk)
                ctx[(int64_t)i * d + h * hd + kk] = spec16 ? bf16_round(oh[i * hd + kk] / rsum[i]) : oh[i * hd + kk];
        free(rsum);
    }
    free(qu); free(qv); free(KT); free(Vh); free(cs); free(ps); free(pr); free(oh); free(cvec);
    linear_t(m->cfg.gemm_bf16, wo, bo, (int)rows, ctx, d, y, d, 0);                  /* :177 */
    for (int64_t i = 0; i < rows * d; ++i) x[i] = x[i] + y[i];     /* :185 */
    free(n); free(q); free(k); free(v); free(ctx); free(y);
    return 0;
}

/* a16: TransformerBlock::forward / TransformerEncoder::forward -- src/transformer.cpp:15-62, :78-88 (the encoder
 * Sortformer stacks on top of the FastConformer; include/parakeet/transformer.hpp:12-21 for the config).
 * Standard multi-head attention: no position term, scale applied to Q K^T (:38), softmax, A V, out_proj; ReLU FFN;
 * pre-LN (x + f(LN(x))) or post-LN (LN(x + f(x))) by `pre_ln`.  Dropout is the identity (inference); the mask branch
 * (:40-42) is not taken by any caller that passes no mask and is not restated.
 * Tensor names: <prefix>layers_.<i>.{norm1_,norm2_}.{weight,bias}, mha_.{q,k,v,out}_proj.{weight,bias},
 * fc1_/fc2_.{weight,bias}, <prefix>final_norm_.{weight,bias} (AX_REGISTER_MODULES order, transformer.cpp:12,69-73).
 * x[B][T][d] in place. */
int orc_transformer_encoder(orc_model *m, const char *prefix, int n_layers, int n_heads, int pre_ln, int has_final_norm,
                            float ln_eps, float *x, int B, int T, int d) {
    const int H = n_heads, hd = d / H;
    if (hd * H != d) return orc_fail("transformer: hidden %d not divisible by %d heads", d, H);
    const float scale = 1.0f / sqrtf((float)hd);                                   /* transformer.cpp:27 */
    for (int l = 0; l < n_layers; ++l) {
        orc_tensor *n1g = getf(m, "%slayers_.%d.norm1_.weight", prefix, l), *n1b = getf(m, "%slayers_.%d.norm1_.bias", prefix, l);
        orc_tensor *n2g = getf(m, "%slayers_.%d.norm2_.weight", prefix, l), *n2b = getf(m, "%slayers_.%d.norm2_.bias", prefix, l);
        orc_tensor *wq = getf(m, "%slayers_.%d.mha_.q_proj.weight", prefix, l), *bq = getf(m, "%slayers_.%d.mha_.q_proj.bias", prefix, l);
        orc_tensor *wk = getf(m, "%slayers_.%d.mha_.k_proj.weight", prefix, l), *bk = getf(m, "%slayers_.%d.mha_.k_proj.bias", prefix, l);
        orc_tensor *wv = getf(m, "%slayers_.%d.mha_.v_proj.weight", prefix, l), *bv = getf(m, "%slayers_.%d.mha_.v_proj.bias", prefix, l);
        orc_tensor *wo = getf(m, "%slayers_.%d.mha_.out_proj.weight", prefix, l), *bo = getf(m, "%slayers_.%d.mha_.out_proj.bias", prefix, l);
        orc_tensor *w1 = getf(m, "%slayers_.%d.fc1_.weight", prefix, l), *b1 = getf(m, "%slayers_.%d.fc1_.bias", prefix, l);
        orc_tensor *w2 = getf(m, "%slayers_.%d.fc2_.weight", prefix, l), *b2 = getf(m, "%slayers_.%d.fc2_.bias", prefix, l);
        if (!n1g || !n1b || !n2g || !n2b || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !w1 || !b1 || !w2 || !b2) return -1;
        const int ffn = (int)w1->shape[0];
        for (int b = 0; b < B; ++b) {
            float *xb = x + (int64_t)b * T * d;
            const int64_t rows = T;
            float *n = (float *)xmalloc((size_t)rows * d * sizeof(float));
            float *q = (float *)xmalloc((size_t)rows * d * sizeof(float));
            float *k = (float *)xmalloc((size_t)rows * d * sizeof(float));
            float *v = (float *)xmalloc((size_t)rows * d * sizeof(float));
            float *ctx = (float *)xmalloc((size_t)rows * d * sizeof(float));
            float *y = (float *)xmalloc((size_t)rows * d * sizeof(float));
            float *hbuf = (float *)xmalloc((size_t)rows * ffn * sizeof(float));
            const float *in = xb;
            if (pre_ln) { layer_norm(xb, rows, d, n1g->data, n1b->data, ln_eps, n); in = n; }       /* :18 */
            linear_t(0, wq, bq, (int)rows, in, d, q, d, 0);                                          /* :20-22 */
            linear_t(0, wk, bk, (int)rows, in, d, k, d, 0);
            linear_t(0, wv, bv, (int)rows, in, d, v, d, 0);
            float *qh = (float *)xmalloc((size_t)T * hd * sizeof(float));
            float *KT = (float *)xmalloc((size_t)hd * T * sizeof(float));
            float *Vh = (float *)xmalloc((size_t)T * hd * sizeof(float));
            float *cs = (float *)xmalloc((size_t)T * T * sizeof(float));
            float *oh = (float *)xmalloc((size_t)T * hd * sizeof(float));
            for (int h = 0; h < H; ++h) {
                for (int i = 0; i < T; ++i)
                    for (int kk = 0; kk < hd; ++kk) {
                        qh[i * hd + kk] = q[(int64_t)i * d + h * hd + kk];
                        KT[(int64_t)kk * T + i] = k[(int64_t)i * d + h * hd + kk];
                        Vh[i * hd + kk] = v[(int64_t)i * d + h * hd + kk];
                    }
                gemm_core(T, T, hd, qh, hd, KT, T, cs, T, 0);                                        /* :38 */
                for (int i = 0; i < T; ++i) {
                    float *row = cs + (int64_t)i * T;
                    float mx = -INFINITY;
                    for (int j = 0; j < T; ++j) {
                        row[j] = row[j] * scale;
                        mx = row[j] > mx ? row[j] : mx;
                    }
                    for (int j = 0; j < T; ++j) row[j] = orc_expf(row[j] - mx);                      /* softmax :44 */
                    const float sum = orc_sum64(row, T, 1);
                    for (int j = 0; j < T; ++j) row[j] = row[j] / sum;
                }
                gemm_core(T, hd, T, cs, T, Vh, hd, oh, hd, 0);                                       /* :45 */
                for (int i = 0; i < T; ++i)
                    for (int kk = 0; kk < hd; ++kk) ctx[(int64_t)i * d + h * hd + kk] = oh[i * hd + kk];
            }
            free(qh); free(KT); free(Vh); free(cs); free(oh);
            linear_t(0, wo, bo, (int)rows, ctx, d, y, d, 0);                                         /* :49 */
            for (int64_t i = 0; i < rows * d; ++i) xb[i] = xb[i] + y[i];                             /* :51 input + out */
            if (!pre_ln) layer_norm(xb, rows, d, n1g->data, n1b->data, ln_eps, xb);
            in = xb;
            if (pre_ln) { layer_norm(xb, rows, d, n2g->data, n2b->data, ln_eps, n); in = n; }       /* :54 */
            linear_t(0, w1, b1, (int)rows, in, d, hbuf, ffn, 0);                                     /* :55 */
            for (int64_t i = 0; i < rows * ffn; ++i) hbuf[i] = hbuf[i] > 0.0f ? hbuf[i] : 0.0f;      /* relu :56 */
            linear_t(0, w2, b2, (int)rows, hbuf, ffn, y, d, 0);                                      /* :58 */
            for (int64_t i = 0; i < rows * d; ++i) xb[i] = xb[i] + y[i];                             /* :61 */
            if (!pre_ln) layer_norm(xb, rows, d, n2g->data, n2b->data, ln_eps, xb);
            free(n); free(q); free(k); free(v); free(ctx); free(y); free(hbuf);
        }
    }
    if (has_final_norm) {                                                                            /* :84-86 */
        orc_tensor *fg = getf(m, "%sfinal_norm_.weight", prefix), *fb = getf(m, "%sfinal_norm_.bias", prefix);
        if (!fg || !fb) return -1;
        layer_norm(x, (int64_t)B * T, d, fg->data, fb->data, ln_eps, x);
    }
    return 0;
}

/* a7: ConformerConvModule::forward -- src/encoder.cpp:59-75.  One utterance: x[T][d]. */
static int conv_module(orc_model *m, int layer, float *x, int T) {
    const orc_config *c = &m->cfg;
    const int d = c->d_model, Kc = c->conv_k, padl = (Kc - 1) / 2;
    const int64_t rows = T;
    orc_tensor *ng = getf(m, "encoder_.layers_.%d.conv_.norm_.weight", layer), *nb = getf(m, "encoder_.layers_.%d.conv_.norm_.bias", layer);
    orc_tensor *w1 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv1_.weight", layer), *b1 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv1_.bias", layer);
    orc_tensor *wd = getf(m, "encoder_.layers_.%d.conv_.depthwise_conv_.weight", layer), *bd = getf(m, "encoder_.layers_.%d.conv_.depthwise_conv_.bias", layer);
    orc_tensor *bng = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.weight", layer), *bnb = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.bias", layer);
    orc_tensor *bnm = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.running_mean", layer), *bnv = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.running_var", layer);
    orc_tensor *w2 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv2_.weight", layer), *b2 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv2_.bias", layer);
    if (!ng || !nb || !w1 || !b1 || !wd || !bd || !bng || !bnb || !bnm || !bnv || !w2 || !b2) return -1;
    float *n = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *g2 = (float *)xmalloc((size_t)rows * 2 * d * sizeof(float));
    float *g = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *dw = (float *)xmalloc((size_t)rows * d * sizeof(float));
    float *y = (float *)xmalloc((size_t)rows * d * sizeof(float));
    layer_norm(x, rows, d, ng->data, nb->data, c->ln_eps, n);                 /* :60 */
    linear_t(m->cfg.gemm_bf16, w1, b1, (int)rows, n, d, g2, 2 * d, 0);                          /* :63 */
    for (int64_t r = 0; r < rows; ++r)                                        /* glu(dim=channels) :64 */
        for (int i = 0; i < d; ++i) g[r * d + i] = g2[r * 2 * d + i] * orc_sigmoidf(g2[r * 2 * d + d + i]);
    for (int t = 0; t < T; ++t)
        for (int ch = 0; ch < d; ++ch) {
            float acc = 0.0f;                                                 /* depthwise k=9 p=4 :66 */
            for (int kk = 0; kk < Kc; ++kk) {
                const int tt = t + kk - padl;
                if (tt < 0 || tt >= T) continue;
                acc = fmaf(wd->data[ch * Kc + kk], g[(int64_t)tt * d + ch], acc);
            }
            float v = acc + bd->data[ch];
            /* BatchNorm1d, inference / running stats (A3, A4) :67 */
            const float rstd = 1.0f / sqrtf(bnv->data[ch] + c->bn_eps);
            v = fmaf((v - bnm->data[ch]) * rstd, bng->data[ch], bnb->data[ch]);
            dw[(int64_t)t * d + ch] = orc_siluf(v);                           /* :68 */
        }
    linear_t(m->cfg.gemm_bf16, w2, b2, (int)rows, dw, d, y, d, 0);                              /* :70 */
    for (int64_t i = 0; i < rows * d; ++i) x[i] = x[i] + y[i];                /* :74 */
    free(n); free(g2); free(g); free(dw); free(y);
    return 0;
}

/* build the lazily cached weight transposes of one layer serially (they are shared by the batch threads) */
static int prepare_layer(orc_model *m, int layer) {
    static const char *names[] = {"ffn1_.fc1_", "ffn1_.fc2_", "ffn2_.fc1_", "ffn2_.fc2_", "attn_.mha_.q_proj",
                                  "attn_.mha_.k_proj", "attn_.mha_.v_proj", "attn_.mha_.out_proj", "attn_.pos_proj_",
                                  "conv_.pointwise_conv1_", "conv_.pointwise_conv2_"};
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); ++i) {
        orc_tensor *t = getf(m, "encoder_.layers_.%d.%s.weight", layer, names[i]);
        if (!t) return -1;
        wt_of(t);
    }
    return 0;
}

/* a8: ConformerBlock::forward -- src/encoder.cpp:196-204.  x[B][T][d] in place.
 * stop_after: 0 = whole block; 1..4 = stop after ffn1 / attn / conv / ffn2 (taps). */
int orc_conformer_block(orc_model *m, int layer, float *x, int B, int T, const float *pos_emb, int stop_after) {
    const orc_config *c = &m->cfg;
    const int d = c->d_model;
    if (prepare_layer(m, layer)) return -1;
    float *PT = pos_proj_heads(m, layer, T, pos_emb, attn_bf16_spec(&m->cfg));
    if (!PT) return -1;
    orc_tensor *g = getf(m, "encoder_.layers_.%d.final_norm_.weight", layer), *bb = getf(m, "encoder_.layers_.%d.final_norm_.bias", layer);
    if (!g || !bb) { free(PT); return -1; }
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
    for (int b = 0; b < B; ++b) {
        float *xb = x + (int64_t)b * T * d;
        int e = feed_forward(m, layer, "ffn1_", xb, T);
        if (!e && stop_after != 1) e = attention(m, layer, xb, T, PT);
        if (!e && stop_after != 1 && stop_after != 2) e = conv_module(m, layer, xb, T);
        if (!e && (stop_after == 0 || stop_after >= 4)) e = feed_forward(m, layer, "ffn2_", xb, T);
        if (!e && stop_after == 0) {
            float *y = (float *)xmalloc((size_t)T * d * sizeof(float));
            layer_norm(xb, T, d, g->data, bb->data, c->ln_eps, y);            /* :202 */
            memcpy(xb, y, (size_t)T * d * sizeof(float));
            free(y);
        }
        err |= (e != 0);
    }
    free(PT);
    return err ? -1 : 0;
}

/* FastConformerEncoder::forward -- src/encoder.cpp:253-271.  Returns T. */
int orc_encoder(orc_model *m, const float *feats, int B, int Tm, float *out, float *layer_taps) {
    const orc_config *c = &m->cfg;
    const int T = orc_subsampling(m, feats, B, Tm, out, NULL, NULL);
    if (T < 0) return -1;
    float *pe = (float *)xmalloc((size_t)(2 * T - 1) * c->d_model * sizeof(float));
    orc_pos_emb(T, c->d_model, pe);
    const int64_t sz = (int64_t)B * T * c->d_model;
    if (m->xscaling) {                                              /* streaming_encoder.cpp:402-406: x = x * sqrt(hidden) */
        const float scale = sqrtf((float)c->d_model);
        for (int64_t i = 0; i < sz; ++i) out[i] = out[i] * scale;
    }
    for (int l = 0; l < c->n_layers; ++l) {
        if (orc_conformer_block(m, l, out, B, T, pe, 0)) { free(pe); return -1; }
        if (layer_taps) memcpy(layer_taps + (int64_t)l * sz, out, (size_t)sz * sizeof(float));
    }
    free(pe);
    return T;
}

/* ------------------------------------------------------------------------- */
/* Sortformer (src/sortformer.cpp:50-121): NEST FastConformer -> projection_ -> TransformerEncoder -> speaker head        */
/* ------------------------------------------------------------------------- */
/* Sortformer::forward (:50-69).  feats[B][Tm][mel] -> probs[B][T][S] (sigmoid speaker activities); returns T.
 * Tensor names (AX_REGISTER_MODULES, :44-46): nest_encoder_.*, projection_, transformer_.*, first_hidden_, output_proj_
 * (hidden_to_spks_ is registered but never used by forward). */
/* projection_ -> transformer_ -> speaker head on encoder frames enc[B][T][d] (src/sortformer.cpp:55-68; :132-141 in diarize_chunk) */
static int sortformer_head(orc_model *m, const float *enc, int B, int T, int n_tlayers, int n_theads, int pre_ln, int has_final_norm, float *probs) {
    const orc_config *c = &m->cfg;
    orc_tensor *pw = getf(m, "projection_.weight"), *pb = getf(m, "projection_.bias");
    orc_tensor *fw = getf(m, "first_hidden_.weight"), *fb = getf(m, "first_hidden_.bias");
    orc_tensor *ow = getf(m, "output_proj_.weight"), *ob = getf(m, "output_proj_.bias");
    if (!pw || !pb || !fw || !fb || !ow || !ob) return -1;
    const int d = c->d_model, dt = (int)pw->shape[0], S = (int)ow->shape[0];
    const int64_t rows = (int64_t)B * T;
    float *x = (float *)xmalloc((size_t)rows * dt * sizeof(float));
    float *h = (float *)xmalloc((size_t)rows * dt * sizeof(float));
    float *lg = (float *)xmalloc((size_t)rows * S * sizeof(float));
    linear_t(0, pw, pb, (int)rows, enc, d, x, dt, 0);                                   /* :55 */
    int r = orc_transformer_encoder(m, "transformer_.", n_tlayers, n_theads, pre_ln, has_final_norm, c->ln_eps, x, B, T, dt);   /* :58 */
    if (r == 0) {
        for (int64_t i = 0; i < rows * dt; ++i) x[i] = x[i] > 0.0f ? x[i] : 0.0f;       /* :62 relu */
        linear_t(0, fw, fb, (int)rows, x, dt, h, dt, 0);                                /* :63 first_hidden_ */
        for (int64_t i = 0; i < rows * dt; ++i) h[i] = h[i] > 0.0f ? h[i] : 0.0f;       /* :64 */
        linear_t(0, ow, ob, (int)rows, h, dt, lg, S, 0);                                /* :65 output_proj_ */
        for (int64_t i = 0; i < rows * S; ++i) probs[i] = orc_sigmoidf(lg[i]);          /* :68 */
    }
    free(x); free(h); free(lg);
    return r;
}
int orc_sortformer_forward(orc_model *m, const float *feats, int B, int Tm, int n_tlayers, int n_theads, int pre_ln,
                           int has_final_norm, float *probs) {
    const orc_config *c = &m->cfg;
    const int d = c->d_model;
    const int T = orc_subsampled_len(Tm);
    float *enc = (float *)xmalloc((size_t)B * T * d * sizeof(float));
    if (orc_encoder(m, feats, B, Tm, enc, NULL) < 0) { free(enc); return -1; }          /* :52 nest_encoder_(features) */
    const int r = sortformer_head(m, enc, B, T, n_tlayers, n_theads, pre_ln, has_final_norm, probs);
    free(enc);
    return r == 0 ? T : -1;
}

/* Sortformer::probs_to_segments (:71-113): per speaker, contiguous runs of prob > threshold -> [start, end] in seconds
 * (frame * 0.08), then sorted by start (std::sort; ties keep no particular order -- here: stable, by speaker).
 * probs[T][S]; out arrays sized >= S * (T + 1) / 2 + S; returns the segment count. */
int orc_probs_to_segments(const float *probs, int T, int S, float threshold, int32_t *spk, float *start, float *end) {
    int n = 0;
    for (int s = 0; s < S; ++s) {
        int in_seg = 0, seg_start = 0;
        for (int t = 0; t < T; ++t) {
            const int active = probs[(int64_t)t * S + s] > threshold;
            if (active && !in_seg) { seg_start = t; in_seg = 1; }
            else if (!active && in_seg) {
                spk[n] = s; start[n] = (float)seg_start * 0.08f; end[n] = (float)(t - 1) * 0.08f; ++n;
                in_seg = 0;
            }
        }
        if (in_seg) { spk[n] = s; start[n] = (float)seg_start * 0.08f; end[n] = (float)(T - 1) * 0.08f; ++n; }
    }
    for (int i = 1; i < n; ++i) {                                  /* insertion sort by start: stable */
        const int32_t ks = spk[i]; const float a = start[i], b = end[i];
        int j = i - 1;
        while (j >= 0 && start[j] > a) { spk[j + 1] = spk[j]; start[j + 1] = start[j]; end[j + 1] = end[j]; --j; }
        spk[j + 1] = ks; start[j + 1] = a; end[j + 1] = b;
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* log_softmax over a row of n logits (max-subtracted; canonical sum64)       */
/* ------------------------------------------------------------------------- */
static void log_softmax_row(const float *x, int n, float *y) {
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = x[i] > mx ? x[i] : mx;
    float tmp[8200];
    float *e = n <= 8200 ? tmp : (float *)xmalloc((size_t)n * sizeof(float));
    for (int i = 0; i < n; ++i) e[i] = orc_expf(x[i] - mx);
    const float lse = orc_logf(orc_sum64(e, n, 1));
    for (int i = 0; i < n; ++i) y[i] = (x[i] - mx) - lse;
    if (e != tmp) free(e);
}
static int argmax_first(const float *x, int n) { /* strict '>' : lowest index wins ties (src/ctc.cpp:59-66) */
    int best = 0;
    float bv = x[0];
    for (int i = 1; i < n; ++i)
        if (x[i] > bv) { bv = x[i]; best = i; }
    return best;
}

/* a9: CTCDecoder::forward -- src/ctc.cpp:12-25.  enc[B][T][d] -> logp[B][T][V] */
int orc_ctc_logprobs(orc_model *m, const float *enc, int B, int T, float *logp) {
    orc_tensor *w = getf(m, "ctc_decoder_.proj_.weight"), *b = getf(m, "ctc_decoder_.proj_.bias");
    if (!w || !b) return -1;
    const int V = (int)w->shape[0], d = m->cfg.d_model;
    const int64_t rows = (int64_t)B * T;
    wt_of(w);
    linear_t(m->cfg.gemm_bf16, w, b, (int)rows, enc, d, logp, V, 1);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float tmp[8200];
        log_softmax_row(logp + r * V, V, tmp);
        memcpy(logp + r * V, tmp, (size_t)V * sizeof(float));
    }
    return V;
}

/* a10: ctc_greedy_decode(+_with_timestamps) -- src/ctc.cpp:40-75, :79-127.
 * ids/start/end/conf are [B][T] (max T tokens per utterance), lens[B]. */
void orc_ctc_greedy(const float *logp, int B, int T, int V, int blank_id, int32_t *ids, int32_t *lens,
                    int32_t *start, int32_t *end, float *conf) {
    for (int b = 0; b < B; ++b) {
        int prev = -1, n = 0;
        for (int t = 0; t < T; ++t) {
            const float *frame = logp + ((int64_t)b * T + t) * V;
            const int best = argmax_first(frame, V);
            if (best != prev) {
                if (prev != -1 && prev != blank_id && n > 0 && end) end[(int64_t)b * T + n - 1] = t - 1; /* :108-110 */
                if (best != blank_id) {
                    ids[(int64_t)b * T + n] = best;
                    if (start) start[(int64_t)b * T + n] = t;
                    if (end) end[(int64_t)b * T + n] = t;
                    if (conf) conf[(int64_t)b * T + n] = orc_expf(frame[best]);                          /* :113 */
                    ++n;
                }
            }
            prev = best;
        }
        if (n > 0 && end) end[(int64_t)b * T + n - 1] = T - 1;                                       /* :121-123 */
        lens[b] = n;
    }
}

/* ------------------------------------------------------------------------- */
/* Phrase boosting -- ContextTrie (src/phrase_boost.cpp:9-66) and the boosted greedy decoders (:70-350)   */
/* ------------------------------------------------------------------------- */
struct orc_trie {
    int n_nodes, cap;
    int *first_child;      /* per node: head of its child list (-1 none) */
    int *next_sibling;     /* per node: next child of the same parent */
    int *token;            /* per node: the token on the edge from its parent */
    int *depth;
};
orc_trie *orc_trie_new(void) {
    orc_trie *t = (orc_trie *)calloc(1, sizeof(*t));
    t->cap = 64;
    t->first_child = (int *)xmalloc(sizeof(int) * t->cap); t->next_sibling = (int *)xmalloc(sizeof(int) * t->cap);
    t->token = (int *)xmalloc(sizeof(int) * t->cap); t->depth = (int *)xmalloc(sizeof(int) * t->cap);
    t->n_nodes = 1; t->first_child[0] = -1; t->next_sibling[0] = -1; t->token[0] = -1; t->depth[0] = 0;   /* root (:9) */
    return t;
}
void orc_trie_free(orc_trie *t) { if (t) { free(t->first_child); free(t->next_sibling); free(t->token); free(t->depth); free(t); } }
int orc_trie_size(const orc_trie *t) { return t->n_nodes; }
static int trie_child(const orc_trie *t, int node, int tok) {
    for (int c = t->first_child[node]; c >= 0; c = t->next_sibling[c])
        if (t->token[c] == tok) return c;
    return -1;
}
void orc_trie_insert(orc_trie *t, const int32_t *ids, int n) {                 /* :11-27 */
    if (n <= 0) return;
    int node = 0;
    for (int i = 0; i < n; ++i) {
        int c = trie_child(t, node, ids[i]);
        if (c < 0) {
            if (t->n_nodes == t->cap) {
                t->cap *= 2;
                t->first_child = (int *)realloc(t->first_child, sizeof(int) * t->cap); t->next_sibling = (int *)realloc(t->next_sibling, sizeof(int) * t->cap);
                t->token = (int *)realloc(t->token, sizeof(int) * t->cap); t->depth = (int *)realloc(t->depth, sizeof(int) * t->cap);
            }
            c = t->n_nodes++;
            t->first_child[c] = -1; t->token[c] = ids[i]; t->depth[c] = t->depth[node] + 1;
            t->next_sibling[c] = t->first_child[node]; t->first_child[node] = c;
        }
        node = c;
    }
}
/* the active-state set: root always in it after an advance (:52-66); at most one state per trie depth */
typedef struct { int n; int s[256]; } trie_active;
static void trie_mark(const orc_trie *t, const trie_active *a, unsigned char *flag, int V, int on) {   /* get_boosted_tokens :39-50 */
    for (int i = 0; i < a->n; ++i)
        for (int c = t->first_child[a->s[i]]; c >= 0; c = t->next_sibling[c])
            if (t->token[c] >= 0 && t->token[c] < V) flag[t->token[c]] = (unsigned char)on;
}
static void trie_advance(const orc_trie *t, trie_active *a, int tok) {
    trie_active nx;
    nx.n = 1; nx.s[0] = 0;
    for (int i = 0; i < a->n; ++i) {
        const int c = trie_child(t, a->s[i], tok);
        if (c >= 0 && nx.n < 256) {
            int dup = 0;
            for (int k = 0; k < nx.n; ++k) dup |= nx.s[k] == c;
            if (!dup) nx.s[nx.n++] = c;
        }
    }
    *a = nx;
}
/* test hooks for the reference's ContextTrie unit tests (tests/test_all.cpp:1280-1353) */
int orc_trie_boosted_tokens(const orc_trie *t, const int32_t *states, int n, unsigned char *flag, int V) {
    trie_active a;
    a.n = 0;
    for (int i = 0; i < n && a.n < 256; ++i)
        if (states[i] >= 0 && states[i] < t->n_nodes) a.s[a.n++] = states[i];       /* out-of-range states are skipped (:43-44) */
    memset(flag, 0, (size_t)V);
    trie_mark(t, &a, flag, V, 1);
    int c = 0;
    for (int v = 0; v < V; ++v) c += flag[v];
    return c;
}
int orc_trie_advance(const orc_trie *t, const int32_t *states, int n, int tok, int32_t *out) {
    trie_active a;
    a.n = 0;
    for (int i = 0; i < n && a.n < 256; ++i)
        if (states[i] >= 0 && states[i] < t->n_nodes) a.s[a.n++] = states[i];
    trie_advance(t, &a, tok);
    for (int i = 0; i < a.n; ++i) out[i] = a.s[i];
    return a.n;
}
static int argmax_boosted(const float *x, int n, const unsigned char *flag, float boost) {   /* strict '>' on value + boost (:88-97) */
    int best = 0;
    float bv = x[0] + (flag[0] ? boost : 0.0f);
    for (int v = 1; v < n; ++v) {
        const float val = x[v] + (flag[v] ? boost : 0.0f);
        if (val > bv) { bv = val; best = v; }
    }
    return best;
}

/* ctc_greedy_decode(_with_timestamps)_boosted -- src/phrase_boost.cpp:70-171.  Same outputs as orc_ctc_greedy; the confidence
 * is exp of the UNBOOSTED log-prob (:151-152). */
void orc_ctc_greedy_boosted(const float *logp, int B, int T, int V, int blank_id, const orc_trie *trie, float boost, int32_t *ids,
                            int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    unsigned char *flag = (unsigned char *)calloc((size_t)V, 1);
    for (int b = 0; b < B; ++b) {
        int prev = -1, n = 0;
        trie_active act;
        act.n = 1; act.s[0] = 0;
        for (int t = 0; t < T; ++t) {
            const float *frame = logp + ((int64_t)b * T + t) * V;
            trie_mark(trie, &act, flag, V, 1);
            const int best = argmax_boosted(frame, V, flag, boost);
            trie_mark(trie, &act, flag, V, 0);
            if (best != prev) {
                if (prev != -1 && prev != blank_id && n > 0 && end) end[(int64_t)b * T + n - 1] = t - 1;
                if (best != blank_id) {
                    ids[(int64_t)b * T + n] = best;
                    if (start) start[(int64_t)b * T + n] = t;
                    if (end) end[(int64_t)b * T + n] = t;
                    if (conf) conf[(int64_t)b * T + n] = orc_expf(frame[best]);
                    ++n;
                    trie_advance(trie, &act, best);                                /* advance on actual emission :103-105 */
                }
            }
            prev = best;
        }
        if (n > 0 && end) end[(int64_t)b * T + n - 1] = T - 1;
        lens[b] = n;
    }
    free(flag);
}

/* ------------------------------------------------------------------------- */
/* a11/a12/a13/a14: prediction net, joint, TDT / RNNT greedy                  */
/* src/rnnt.cpp:22-28,37-44,56-111 ; src/lstm.cpp:11-49 ; src/tdt.cpp:15-24,36-201 */
/* ------------------------------------------------------------------------- */
typedef struct {
    int L, Hp, J, V, D, de;
    int bf16;   /* gemm_bf16 mode: the decode products take bf16 weights and bf16-stored h' / z (kernels/decode_gemv_bf16.hip); the layer-0
                 * input projection stays fp32 (the device reads it from the fp32 table g1 = W_ih0 E + b) */
    orc_tensor *embed, *wih[4], *bih[4], *whh[4];
    orc_tensor *we, *be, *wp, *bp, *wl, *bl, *wd, *bd;
} dec_weights;

static int dec_weights_get(orc_model *m, dec_weights *w, int rnnt) {
    const orc_config *c = &m->cfg;
    memset(w, 0, sizeof(*w));
    w->L = c->lstm_layers; w->Hp = c->pred_hidden; w->J = c->joint_hidden; w->V = c->vocab; w->D = c->n_durations; w->de = c->d_model;
    if (w->L > 4) return orc_fail("lstm_layers > 4");
    w->embed = getf(m, "prediction_.embed_.weight");
    if (!w->embed) return -1;
    for (int l = 0; l < w->L; ++l) {
        w->wih[l] = getf(m, "prediction_.lstm_.cells_.%d.input_proj_.weight", l);
        w->bih[l] = getf(m, "prediction_.lstm_.cells_.%d.input_proj_.bias", l);
        w->whh[l] = getf(m, "prediction_.lstm_.cells_.%d.hidden_proj_.weight", l);
        if (!w->wih[l] || !w->bih[l] || !w->whh[l]) return -1;
    }
    const char *jp = c->joint_prefix;
    w->we = getf(m, "%senc_proj_.weight", jp); w->be = getf(m, "%senc_proj_.bias", jp);
    w->wp = getf(m, "%spred_proj_.weight", jp);
    if (!w->we || !w->be || !w->wp) return -1;
    if (c->joint_pred_bias) { /* switch A5: NeMo adds it, the reference's Linear(bias=false) drops it */
        char nm[192];
        snprintf(nm, sizeof nm, "%spred_proj_.bias", jp);
        w->bp = find(m, nm);
    }
    if (rnnt) {
        w->wl = getf(m, "%sout_proj_.weight", jp); w->bl = getf(m, "%sout_proj_.bias", jp);
        if (!w->wl || !w->bl) return -1;
    } else {
        w->wl = getf(m, "%slabel_proj_.weight", jp); w->bl = getf(m, "%slabel_proj_.bias", jp);
        w->wd = getf(m, "%sduration_proj_.weight", jp); w->bd = getf(m, "%sduration_proj_.bias", jp);
        if (!w->wl || !w->bl || !w->wd || !w->bd) return -1;
    }
    /* build transposes up front (thread-safety of the lazy cache) */
    for (int l = 0; l < w->L; ++l) { wt_of(w->wih[l]); wt_of(w->whh[l]); }
    wt_of(w->we); wt_of(w->wp); wt_of(w->wl);
    if (w->wd) wt_of(w->wd);
    w->bf16 = c->gemm_bf16 && (w->Hp % 32 == 0) && (w->J % 32 == 0);
    if (w->bf16) {
        for (int l = 0; l < w->L; ++l) { wt16_of(w->wih[l]); wt16_of(w->whh[l]); }
        wt16_of(w->wp); wt16_of(w->wl);
        if (w->wd) wt16_of(w->wd);
    }
    return 0;
}

/* RNNTPrediction::step (src/rnnt.cpp:22-28) + LSTM::step (src/lstm.cpp:40-49) + LSTMCell::forward (:11-29) */
static void predict_step(const dec_weights *w, int token, float *h /*[L][Hp]*/, float *c /*[L][Hp]*/, float *pred, float *scratch) {
    const int Hp = w->Hp, G = 4 * Hp;
    float *gi = scratch, *gh = scratch + G;
    const float *in = w->embed->data + (int64_t)token * Hp; /* Embedding lookup; blank row is zeros by training (src/tdt.cpp:56-57) */
    for (int l = 0; l < w->L; ++l) {
        /* bf16 mode: h is stored rounded (below), so the operands are already bf16 values; only the weights change */
        gemm_core(1, G, Hp, in, Hp, (w->bf16 && l > 0) ? w->wih[l]->wt16 : w->wih[l]->wt, G, gi, G, 0);
        gemm_core(1, G, Hp, h + l * Hp, Hp, w->bf16 ? w->whh[l]->wt16 : w->whh[l]->wt, G, gh, G, 0);
        for (int j = 0; j < Hp; ++j) {
            /* gates = input_proj(x) + hidden_proj(h); chunk(4): i, f, g, o */
            const float gi_ = (gi[j] + w->bih[l]->data[j]) + gh[j];
            const float gf_ = (gi[Hp + j] + w->bih[l]->data[Hp + j]) + gh[Hp + j];
            const float gg_ = (gi[2 * Hp + j] + w->bih[l]->data[2 * Hp + j]) + gh[2 * Hp + j];
            const float go_ = (gi[3 * Hp + j] + w->bih[l]->data[3 * Hp + j]) + gh[3 * Hp + j];
            const float ig = orc_sigmoidf(gi_), fg = orc_sigmoidf(gf_), gg = orc_tanhf(gg_), og = orc_sigmoidf(go_);
            const float cn = fg * c[l * Hp + j] + ig * gg;          /* c_new = f*c + i*g   (separate mul, add) */
            c[l * Hp + j] = cn;
            h[l * Hp + j] = w->bf16 ? bf16_round(og * orc_tanhf(cn)) : og * orc_tanhf(cn);
        }
        in = h + l * Hp;
    }
    memcpy(pred, h + (w->L - 1) * Hp, (size_t)Hp * sizeof(float));
}

/* TDTJoint::forward (src/tdt.cpp:15-24) on a pre-projected encoder frame ep = enc_proj(enc_t) (bias included) */
static void joint_hidden(const dec_weights *w, const float *ep, const float *pred, float *z, float *scratch) {
    const int J = w->J;
    gemm_core(1, J, w->Hp, pred, w->Hp, w->bf16 ? w->wp->wt16 : w->wp->wt, J, scratch, J, 0);
    for (int j = 0; j < J; ++j) {
        float pj = scratch[j];
        if (w->bp) pj = pj + w->bp->data[j];
        const float s = ep[j] + pj;
        z[j] = s > 0.0f ? s : 0.0f;
        if (w->bf16) z[j] = bf16_round(z[j]);                      /* z is stored as bf16: the heads' operand */
    }
}

/*
 * tdt_greedy_decode(+_with_timestamps): src/tdt.cpp:36-110, :122-201.
 * enc[B][T][d].  Outputs [B][max_tokens]; lens[B]; steps[B] = joint evaluations.
 * max_steps: safety cap the reference does not have (its loop can spin forever if
 * 10 duration-0 emissions repeat on one frame); exceeding it returns 1 and sets
 * lens[b] = -1 for that utterance.
 */
static int tdt_greedy_ex(orc_model *m, const float *enc, int B, int T, int max_tokens, int max_steps, int32_t *ids,
                         int32_t *lens, int32_t *start, int32_t *end, float *conf, int32_t *steps,
                         float *first_label_logp /* optional [B][V]: label log-probs of the first joint call */,
                         float *state_hc /* optional [B][2][L][Hp] carried LSTM state (in/out); NULL: zeros */,
                         int32_t *state_token /* optional [B] carried last token (in/out); NULL: blank */, int clamp_end,
                         const orc_trie *trie /* optional phrase-boost trie (src/phrase_boost.cpp:177-350) */, float boost,
                         float *min_margin /* optional [B]: smallest top-1 minus top-2 log-prob (label head and, for TDT, duration head) over the utterance's decisions */,
                         float *step_margin /* optional [B][step_cap]: that margin of EVERY decision (joint evaluation), in order */,
                         int32_t *step_label /* optional [B][step_cap]: the label each decision chose (blank included) */, int step_cap) {
    const orc_config *c = &m->cfg;
    dec_weights w;
    if (dec_weights_get(m, &w, 0)) return -1;
    const int Hp = w.Hp, J = w.J, V = w.V, D = w.D, d = c->d_model;
    /* enc_proj hoisted: one GEMM for all frames (bit-identical to the per-frame call) */
    float *ep = (float *)xmalloc((size_t)B * T * J * sizeof(float));
    linear_t(m->cfg.gemm_bf16, w.we, w.be, B * T, enc, d, ep, J, 1);
    int overflow = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : overflow)
    for (int b = 0; b < B; ++b) {
        float *h = (float *)calloc((size_t)w.L * Hp * 2, sizeof(float)), *cc = h + w.L * Hp;
        if (state_hc) memcpy(h, state_hc + (int64_t)b * 2 * w.L * Hp, (size_t)w.L * Hp * 2 * sizeof(float));
        float *sh = (float *)xmalloc((size_t)w.L * Hp * 2 * sizeof(float));
        float *pred = (float *)xmalloc((size_t)Hp * sizeof(float));
        float *z = (float *)xmalloc((size_t)J * sizeof(float));
        float *scratch = (float *)xmalloc((size_t)(8 * Hp + J) * sizeof(float));
        float *lab = (float *)xmalloc((size_t)V * 2 * sizeof(float)), *lab_lp = lab + V;
        float dur[16], dur_lp[16];
        unsigned char *flag = trie ? (unsigned char *)calloc((size_t)V, 1) : NULL;
        trie_active act;
        act.n = 1; act.s[0] = 0;
        int token = state_token ? state_token[b] : c->blank_id, t = 0, n = 0, nsteps = 0, bad = 0;
        float margin = HUGE_VALF;
        while (t < T && !bad) {
            const float *ept = ep + ((int64_t)b * T + t) * J;
            for (int sym = 0; sym < c->max_symbols; ++sym) {
                if (max_steps > 0 && nsteps >= max_steps) { bad = 1; break; }
                memcpy(sh, h, (size_t)w.L * Hp * 2 * sizeof(float));      /* saved_states = states  :70 */
                predict_step(&w, token, h, cc, pred, scratch);
                joint_hidden(&w, ept, pred, z, scratch);
                gemm_core(1, V, J, z, J, w.bf16 ? w.wl->wt16 : w.wl->wt, V, lab, V, 0);
                for (int i = 0; i < V; ++i) lab[i] = lab[i] + w.bl->data[i];
                log_softmax_row(lab, V, lab_lp);
                gemm_core(1, D, J, z, J, w.bf16 ? w.wd->wt16 : w.wd->wt, D, dur, D, 0);
                for (int i = 0; i < D; ++i) dur[i] = dur[i] + w.bd->data[i];
                log_softmax_row(dur, D, dur_lp);
                if (nsteps == 0 && first_label_logp) memcpy(first_label_logp + (int64_t)b * V, lab_lp, (size_t)V * sizeof(float));
                ++nsteps;
                int k;
                if (trie) {                                                /* argmax of log-prob + boost for the trie's next tokens (phrase_boost.cpp:301-312) */
                    trie_mark(trie, &act, flag, V, 1);
                    k = argmax_boosted(lab_lp, V, flag, boost);
                    trie_mark(trie, &act, flag, V, 0);
                } else {
                    k = argmax_first(lab_lp, V);                           /* :78-82 */
                    if (min_margin) {                                      /* how close the decision was to flipping */
                        float second = -HUGE_VALF;
                        for (int i = 0; i < V; ++i) if (i != k && lab_lp[i] > second) second = lab_lp[i];
                        float mg = lab_lp[k] - second;
                        if (D > 1) {                                       /* the duration argmax is a decision too: a flip there moves the frame pointer */
                            const int dbest = argmax_first(dur_lp, D);
                            float dsecond = -HUGE_VALF;
                            for (int i = 0; i < D; ++i) if (i != dbest && dur_lp[i] > dsecond) dsecond = dur_lp[i];
                            const float dmg = dur_lp[dbest] - dsecond;
                            if (dmg < mg) mg = dmg;
                        }
                        if (mg < margin) margin = mg;
                        if (step_margin && nsteps - 1 < step_cap) step_margin[(int64_t)b * step_cap + nsteps - 1] = mg;
                        if (step_label && nsteps - 1 < step_cap) step_label[(int64_t)b * step_cap + nsteps - 1] = k;
                    }
                }
                const int di = argmax_first(dur_lp, D);
                const int skip = di < D ? c->durations[di] : 1;            /* :84-86 */
                if (k == c->blank_id) {
                    memcpy(h, sh, (size_t)w.L * Hp * 2 * sizeof(float));  /* states = saved_states :90 */
                    t += skip > 1 ? skip : 1;
                    break;
                }
                if (n < max_tokens) {
                    ids[(int64_t)b * max_tokens + n] = k;
                    if (start) start[(int64_t)b * max_tokens + n] = t;
                    if (end) {
                        int e = t + (skip > 1 ? skip : 1) - 1;                /* :184-187 */
                        if (clamp_end && e >= T) e = T - 1;                   /* the streaming decoder does not clamp (eou.cpp:78-79) */
                        end[(int64_t)b * max_tokens + n] = e;
                    }
                    if (conf) conf[(int64_t)b * max_tokens + n] = orc_expf(lab_lp[k]); /* :169 */
                }
                ++n;
                token = k;
                if (trie) trie_advance(trie, &act, k);                     /* phrase_boost.cpp:336 */
                if (skip > 0) { t += skip; break; }
                /* skip == 0: stay on this frame; if the for runs out t is NOT advanced (:66,99-105) */
            }
        }
        lens[b] = bad ? -1 : (n < max_tokens ? n : max_tokens);
        if (steps) steps[b] = nsteps;
        if (min_margin) min_margin[b] = margin;
        overflow |= bad;
        if (state_hc) memcpy(state_hc + (int64_t)b * 2 * w.L * Hp, h, (size_t)w.L * Hp * 2 * sizeof(float));
        if (state_token) state_token[b] = token;
        free(flag);
        free(h); free(sh); free(pred); free(z); free(scratch); free(lab);
    }
    free(ep);
    return overflow;
}
int orc_tdt_greedy(orc_model *m, const float *enc, int B, int T, int max_tokens, int max_steps, int32_t *ids,
                   int32_t *lens, int32_t *start, int32_t *end, float *conf, int32_t *steps, float *first_label_logp) {
    return tdt_greedy_ex(m, enc, B, T, max_tokens, max_steps, ids, lens, start, end, conf, steps, first_label_logp, NULL, NULL, 1, NULL, 0.0f, NULL, NULL, NULL, 0);
}
/* the same decode, also reporting per utterance the smallest top-1 / top-2 label log-prob margin of its decisions */
int orc_tdt_greedy_margin(orc_model *m, const float *enc, int B, int T, int max_tokens, int max_steps, int32_t *ids,
                          int32_t *lens, int32_t *start, int32_t *end, float *conf, int32_t *steps, float *min_margin,
                          float *step_margin, int32_t *step_label, int step_cap) {
    return tdt_greedy_ex(m, enc, B, T, max_tokens, max_steps, ids, lens, start, end, conf, steps, NULL, NULL, NULL, 1, NULL, 0.0f, min_margin,
                         step_margin, step_label, step_cap);
}
/* tdt_greedy_decode(_with_timestamps)_boosted -- src/phrase_boost.cpp:177-350 (confidence = exp of the unboosted log-prob, :313-315) */
int orc_tdt_greedy_boosted(orc_model *m, const float *enc, int B, int T, int max_tokens, int max_steps, const orc_trie *trie, float boost,
                           int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf, int32_t *steps) {
    return tdt_greedy_ex(m, enc, B, T, max_tokens, max_steps, ids, lens, start, end, conf, steps, NULL, NULL, NULL, 1, trie, boost, NULL, NULL, NULL, 0);
}

/*
 * Teacher-forced joint scores: the loop of tdt_greedy_decode (src/tdt.cpp:62-106) on ONE utterance enc[T][d] with the decision of every
 * step GIVEN -- labels_in[k], dur_in[k] (an index into the duration table) -- instead of taken from the argmax; labels_in == NULL: the
 * decisions are the greedy ones (the function then IS the greedy decode) and are reported in labels_out / dur_out.  After step k the frame
 * pointer, the last token and the LSTM state are what the reference's loop holds after deciding that way: a blank reverts the state and
 * advances by max(duration, 1) (:88-93), a token commits it and advances by its duration, 0 = stay on the frame (:95-105).
 * label_lp[k][V], dur_lp[k][D] (either may be NULL) = the joint's log-softmax outputs of step k (TDTJoint::forward, :15-24).
 * Returns the number of steps evaluated (<= n_steps: the walk ends when the frame pointer leaves the utterance), -1 on error.
 */
static int tdt_score_ex(orc_model *m, const float *enc, int T, const int32_t *labels_in, const int32_t *dur_in, int n_steps, int32_t *labels_out,
                        int32_t *dur_out, float *label_lp, float *dur_lp,
                        float *state_hc /* optional [2][L][Hp] carried LSTM state (in/out); NULL: zeros */,
                        int32_t *state_token /* optional carried last token (in/out); NULL: blank */) {
    const orc_config *c = &m->cfg;
    dec_weights w;
    if (dec_weights_get(m, &w, 0)) return -1;
    const int Hp = w.Hp, J = w.J, V = w.V, D = w.D, d = c->d_model;
    if (D < 1 || D > 16) return orc_fail("orc_tdt_score: a TDT joint (1..16 durations) is required");
    float *ep = (float *)xmalloc((size_t)T * J * sizeof(float));
    linear_t(m->cfg.gemm_bf16, w.we, w.be, T, enc, d, ep, J, 1);
    float *h = (float *)calloc((size_t)w.L * Hp * 2, sizeof(float)), *cc = h + w.L * Hp;
    if (state_hc) memcpy(h, state_hc, (size_t)w.L * Hp * 2 * sizeof(float));
    float *sh = (float *)xmalloc((size_t)w.L * Hp * 2 * sizeof(float));
    float *pred = (float *)xmalloc((size_t)Hp * sizeof(float));
    float *z = (float *)xmalloc((size_t)J * sizeof(float));
    float *scratch = (float *)xmalloc((size_t)(8 * Hp + J) * sizeof(float));
    float *lab = (float *)xmalloc((size_t)V * 2 * sizeof(float)), *lab_lp = lab + V;
    float dur[16], dlp[16];
    int token = state_token ? *state_token : c->blank_id, t = 0, k = 0, bad = 0;
    for (; k < n_steps && t < T; ++k) {
        memcpy(sh, h, (size_t)w.L * Hp * 2 * sizeof(float));
        predict_step(&w, token, h, cc, pred, scratch);
        joint_hidden(&w, ep + (int64_t)t * J, pred, z, scratch);
        gemm_core(1, V, J, z, J, w.bf16 ? w.wl->wt16 : w.wl->wt, V, lab, V, 0);
        for (int i = 0; i < V; ++i) lab[i] = lab[i] + w.bl->data[i];
        log_softmax_row(lab, V, lab_lp);
        gemm_core(1, D, J, z, J, w.bf16 ? w.wd->wt16 : w.wd->wt, D, dur, D, 0);
        for (int i = 0; i < D; ++i) dur[i] = dur[i] + w.bd->data[i];
        log_softmax_row(dur, D, dlp);
        if (label_lp) memcpy(label_lp + (int64_t)k * V, lab_lp, (size_t)V * sizeof(float));
        if (dur_lp) memcpy(dur_lp + (int64_t)k * D, dlp, (size_t)D * sizeof(float));
        const int lk = labels_in ? labels_in[k] : argmax_first(lab_lp, V);
        const int di = labels_in ? dur_in[k] : argmax_first(dlp, D);
        if (lk < 0 || lk >= V || di < 0 || di >= D) { bad = 1; break; }
        if (labels_out) labels_out[k] = lk;
        if (dur_out) dur_out[k] = di;
        const int skip = c->durations[di];
        if (lk == c->blank_id) {
            memcpy(h, sh, (size_t)w.L * Hp * 2 * sizeof(float));
            t += skip > 1 ? skip : 1;
        } else {
            token = lk;
            if (skip > 0) t += skip;
        }
    }
    if (state_hc && !bad) memcpy(state_hc, h, (size_t)w.L * Hp * 2 * sizeof(float));
    if (state_token && !bad) *state_token = token;
    free(ep); free(h); free(sh); free(pred); free(z); free(scratch); free(lab);
    if (bad) return orc_fail("orc_tdt_score: forced label / duration index out of range at step %d", k);
    return k;
}
int orc_tdt_score(orc_model *m, const float *enc, int T, const int32_t *labels_in, const int32_t *dur_in, int n_steps, int32_t *labels_out,
                  int32_t *dur_out, float *label_lp, float *dur_lp) {
    return tdt_score_ex(m, enc, T, labels_in, dur_in, n_steps, labels_out, dur_out, label_lp, dur_lp, NULL, NULL);
}

/* rnnt_greedy_decode(+_with_timestamps): src/rnnt.cpp:56-111, :115-177 ; RNNTJoint::forward :37-44 */
int orc_rnnt_greedy(orc_model *m, const float *enc, int B, int T, int max_tokens, int32_t *ids, int32_t *lens,
                    int32_t *start, float *conf) {
    const orc_config *c = &m->cfg;
    dec_weights w;
    if (dec_weights_get(m, &w, 1)) return -1;
    const int Hp = w.Hp, J = w.J, V = w.V, d = c->d_model;
    float *ep = (float *)xmalloc((size_t)B * T * J * sizeof(float));
    linear_t(m->cfg.gemm_bf16, w.we, w.be, B * T, enc, d, ep, J, 1);
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        float *h = (float *)calloc((size_t)w.L * Hp * 2, sizeof(float)), *cc = h + w.L * Hp;
        float *sh = (float *)xmalloc((size_t)w.L * Hp * 2 * sizeof(float));
        float *pred = (float *)xmalloc((size_t)Hp * sizeof(float));
        float *z = (float *)xmalloc((size_t)J * sizeof(float));
        float *scratch = (float *)xmalloc((size_t)(8 * Hp + J) * sizeof(float));
        float *lab = (float *)xmalloc((size_t)V * 2 * sizeof(float)), *lab_lp = lab + V;
        int token = c->blank_id, n = 0;
        for (int t = 0; t < T; ++t) {
            const float *ept = ep + ((int64_t)b * T + t) * J;
            for (int sym = 0; sym < c->max_symbols; ++sym) {
                memcpy(sh, h, (size_t)w.L * Hp * 2 * sizeof(float));
                predict_step(&w, token, h, cc, pred, scratch);
                joint_hidden(&w, ept, pred, z, scratch);
                gemm_core(1, V, J, z, J, w.bf16 ? w.wl->wt16 : w.wl->wt, V, lab, V, 0);
                for (int i = 0; i < V; ++i) lab[i] = lab[i] + w.bl->data[i];
                log_softmax_row(lab, V, lab_lp);
                const int k = argmax_first(lab_lp, V);
                if (k == c->blank_id) { memcpy(h, sh, (size_t)w.L * Hp * 2 * sizeof(float)); break; }
                if (n < max_tokens) {
                    ids[(int64_t)b * max_tokens + n] = k;
                    if (start) start[(int64_t)b * max_tokens + n] = t;
                    if (conf) conf[(int64_t)b * max_tokens + n] = orc_expf(lab_lp[k]);
                }
                ++n;
                token = k;
            }
        }
        lens[b] = n < max_tokens ? n : max_tokens;
        free(h); free(sh); free(pred); free(z); free(scratch); free(lab);
    }
    free(ep);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Streaming path (BASELINE configs[4]): one stream = the state of                     */
/* StreamingAudioPreprocessor (src/audio.cpp:171-259), EncoderCache / BlockCache                       */
/* (include/parakeet/streaming_encoder.hpp:25-41), StreamingDecodeState (include/parakeet/eou.hpp:80-87) */
/* Tolerance-class mode (orc_config.gemm_bf16) of the streaming path -- the specification of the product's  */
/* bf16 streaming mode (kernels/gemm_smallm_bf16.hip): EVERY Linear / 1x1-conv product of the chunk (sub-   */
/* sampling, ffn, q / k / v / out, pointwise convs, pos_proj, the joint's enc_proj and the decode products)  */
/* takes both operands rounded to bf16 with fp32 accumulation; everything between the products (LayerNorm,   */
/* the cached attention incl. its position table, depthwise conv, caches) stays fp32 arithmetic on those      */
/* outputs -- the caches hold fp32 values, nothing else is stored rounded.                                    */
/* ------------------------------------------------------------------------- */
struct orc_stream {
    orc_model *m;
    int att_left, att_right, n_mels;
    /* StreamingAudioPreprocessor */
    float preemph_last;
    float *overlap; int n_overlap;
    /* CausalConvSubsampling::forward_cached: leftover mel frames (< 8) */
    float *mel_cache; int n_mel_cache;
    /* per layer: key / value cache [n_kv][d] (n_kv <= att_left), conv cache [Kc-1][d] (has_conv: first chunk zero-pads) */
    float **kc, **vc; int *n_kv; float **cc; int *has_conv;
    /* StreamingDecodeState */
    float *hc; int32_t token; int frame_offset; int dec_init;
};

orc_stream *orc_stream_new(orc_model *m, int att_left, int att_right) {
    const orc_config *c = &m->cfg;
    orc_stream *s = (orc_stream *)calloc(1, sizeof(*s));
    s->m = m; s->att_left = att_left; s->att_right = att_right; s->n_mels = c->mel_bins;
    s->overlap = (float *)xmalloc(sizeof(float));
    s->mel_cache = (float *)xmalloc((size_t)8 * c->mel_bins * sizeof(float));
    const int L = c->n_layers;
    s->kc = (float **)calloc((size_t)L, sizeof(float *)); s->vc = (float **)calloc((size_t)L, sizeof(float *));
    s->cc = (float **)calloc((size_t)L, sizeof(float *));
    s->n_kv = (int *)calloc((size_t)L, sizeof(int)); s->has_conv = (int *)calloc((size_t)L, sizeof(int));
    for (int l = 0; l < L; ++l) {
        s->kc[l] = (float *)xmalloc((size_t)(att_left > 0 ? att_left : 1) * c->d_model * sizeof(float));
        s->vc[l] = (float *)xmalloc((size_t)(att_left > 0 ? att_left : 1) * c->d_model * sizeof(float));
        s->cc[l] = (float *)xmalloc((size_t)c->conv_k * c->d_model * sizeof(float));
    }
    s->hc = (float *)calloc((size_t)2 * c->lstm_layers * c->pred_hidden, sizeof(float));
    s->token = c->blank_id;
    return s;
}
void orc_stream_free(orc_stream *s) {
    if (!s) return;
    for (int l = 0; l < s->m->cfg.n_layers; ++l) { free(s->kc[l]); free(s->vc[l]); free(s->cc[l]); }
    free(s->kc); free(s->vc); free(s->cc); free(s->n_kv); free(s->has_conv);
    free(s->overlap); free(s->mel_cache); free(s->hc); free(s);
}

/* StreamingAudioPreprocessor::process_chunk -- src/audio.cpp:195-259.  pcm[n] -> out[n_frames][n_mels] (log-mel, NOT
 * normalised); returns n_frames (0: everything was buffered).  out must hold (n_overlap + n) / 160 + 1 frames.
 * center=false: a frame is win_length (400) samples -- the reference's own frame count, (total - 400) / 160 + 1 (:222-223),
 * only makes sense that way -- Hann-windowed and zero-padded on the right to the 512-point FFT (the left-aligned placement
 * of switch A1; the centred alternative does not exist for this path).  Quirk kept: the next chunk's first frame starts
 * AFTER the last window (consumed = (n_frames-1)*160 + 400, :230-231), not one hop after the last frame start. */
int orc_stream_mel(orc_stream *s, const float *pcm, int n, float *out) {
    const int n_fft = 512, win = 400, hop = 160, n_freqs = 257, n_mels = s->n_mels;
    const int total = s->n_overlap + n;
    float *buf = (float *)xmalloc((size_t)(total > 0 ? total : 1) * sizeof(float));
    memcpy(buf, s->overlap, (size_t)s->n_overlap * sizeof(float));
    for (int i = 0; i < n; ++i) {                                              /* 1. preemphasis with carried sample :204-210 */
        const float cur = pcm[i];
        const float t = 0.97f * s->preemph_last;
        buf[s->n_overlap + i] = cur - t;
        s->preemph_last = cur;
    }
    int n_frames = total < win ? 0 : (total - win) / hop + 1;
    if (n_frames <= 0) {                                                       /* :216-228 buffer everything */
        free(s->overlap); s->overlap = buf; s->n_overlap = total;
        return 0;
    }
    const int consumed = (n_frames - 1) * hop + win;
    float *w = (float *)xmalloc((size_t)win * sizeof(float));
    for (int k = 0; k < win; ++k) w[k] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)k / (double)(win - 1)));
    float *twr = (float *)xmalloc((size_t)n_fft / 2 * sizeof(float)), *twi = (float *)xmalloc((size_t)n_fft / 2 * sizeof(float));
    fft_twiddles(n_fft, twr, twi);
    float *fb = (float *)xmalloc((size_t)n_freqs * n_mels * sizeof(float));
    orc_mel_filterbank(n_freqs, n_mels, 16000.0f, 0.0f, 8000.0f, fb);
    float re[512], im[512], pw[257];
    for (int t = 0; t < n_frames; ++t) {
        for (int k = 0; k < n_fft; ++k) {
            re[k] = k < win ? buf[t * hop + k] * w[k] : 0.0f;
            im[k] = 0.0f;
        }
        fft_radix2(n_fft, 9, re, im, twr, twi);
        for (int f = 0; f < n_freqs; ++f) {
            const float q = fmaf(re[f], re[f], im[f] * im[f]);
            const float mag = sqrtf(q);                                        /* abs() then square :243-244 */
            pw[f] = mag * mag;
        }
        for (int mm = 0; mm < n_mels; ++mm) {
            float acc = 0.0f;
            for (int f = 0; f < n_freqs; ++f) acc = fmaf(fb[f * n_mels + mm], pw[f], acc);
            out[(int64_t)t * n_mels + mm] = orc_logf(acc + 5.96046448e-8f);    /* :251-252 ; no normalisation */
        }
    }
    const int rest = total - consumed;
    float *ov = (float *)xmalloc((size_t)(rest > 0 ? rest : 1) * sizeof(float));
    memcpy(ov, buf + consumed, (size_t)rest * sizeof(float));
    free(s->overlap); s->overlap = ov; s->n_overlap = rest;
    free(buf); free(w); free(twr); free(twi); free(fb);
    return n_frames;
}

/* StreamingConformerAttention::forward_cached -- src/streaming_encoder.cpp:162-272.  x[c][d] in place.
 * Quirks kept literally: the position scores are NOT rel-shifted here -- the rightmost kv_len columns of (q+v) P^T are
 * added to the content scores (:215-224); the K/V cache is trimmed to att_left rows AFTER this chunk was appended, yet the
 * scores use the untrimmed kv (:186-203); masked entries are REPLACED by -1e9 (:226-247). */
static int stream_attention(orc_stream *s, int layer, float *x, int c, const float *PT /*[H][hd][P]*/, int P) {
    orc_model *m = s->m;
    const orc_config *cf = &m->cfg;
    const int d = cf->d_model, H = cf->n_heads, hd = d / H;
    orc_tensor *ng = getf(m, "encoder_.layers_.%d.attn_.norm_.weight", layer), *nb = getf(m, "encoder_.layers_.%d.attn_.norm_.bias", layer);
    orc_tensor *wq = getf(m, "encoder_.layers_.%d.attn_.mha_.q_proj.weight", layer), *bq = getf(m, "encoder_.layers_.%d.attn_.mha_.q_proj.bias", layer);
    orc_tensor *wk = getf(m, "encoder_.layers_.%d.attn_.mha_.k_proj.weight", layer), *bk = getf(m, "encoder_.layers_.%d.attn_.mha_.k_proj.bias", layer);
    orc_tensor *wv = getf(m, "encoder_.layers_.%d.attn_.mha_.v_proj.weight", layer), *bv = getf(m, "encoder_.layers_.%d.attn_.mha_.v_proj.bias", layer);
    orc_tensor *wo = getf(m, "encoder_.layers_.%d.attn_.mha_.out_proj.weight", layer), *bo = getf(m, "encoder_.layers_.%d.attn_.mha_.out_proj.bias", layer);
    orc_tensor *pu = getf(m, "encoder_.layers_.%d.attn_.pos_bias_u_", layer), *pv = getf(m, "encoder_.layers_.%d.attn_.pos_bias_v_", layer);
    if (!ng || !nb || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !pu || !pv) return -1;
    const int nc = s->n_kv[layer], kv = nc + c;
    float *n = (float *)xmalloc((size_t)c * d * sizeof(float)), *q = (float *)xmalloc((size_t)c * d * sizeof(float));
    float *k = (float *)xmalloc((size_t)kv * d * sizeof(float)), *v = (float *)xmalloc((size_t)kv * d * sizeof(float));
    float *ctx = (float *)xmalloc((size_t)c * d * sizeof(float)), *y = (float *)xmalloc((size_t)c * d * sizeof(float));
    layer_norm(x, c, d, ng->data, nb->data, cf->ln_eps, n);                          /* :165 */
    linear_t(cf->gemm_bf16, wq, bq, c, n, d, q, d, 0);                                           /* :168-170 */
    memcpy(k, s->kc[layer], (size_t)nc * d * sizeof(float));                         /* prepend the cache :186-189 */
    memcpy(v, s->vc[layer], (size_t)nc * d * sizeof(float));
    linear_t(cf->gemm_bf16, wk, bk, c, n, d, k + (int64_t)nc * d, d, 0);
    linear_t(cf->gemm_bf16, wv, bv, c, n, d, v + (int64_t)nc * d, d, 0);
    {   /* cache <- last att_left rows of kv :193-209 */
        const int keep = kv > s->att_left ? s->att_left : kv, from = kv - keep;
        memmove(s->kc[layer], k + (int64_t)from * d, (size_t)keep * d * sizeof(float));
        memmove(s->vc[layer], v + (int64_t)from * d, (size_t)keep * d * sizeof(float));
        s->n_kv[layer] = keep;
    }
    const float scale = 1.0f / sqrtf((float)hd);
    const int off = P > kv ? P - kv : 0;                                             /* rightmost kv columns :217-224 */
    float *row = (float *)xmalloc((size_t)kv * sizeof(float));
    for (int h = 0; h < H; ++h)
        for (int i = 0; i < c; ++i) {
            const int abs_pos = kv - c + i;
            float mx = -INFINITY;
            for (int j = 0; j < kv; ++j) {
                float cs = 0.0f, ps = 0.0f;
                for (int e = 0; e < hd; ++e) {
                    const float qq = q[(int64_t)i * d + h * hd + e];
                    cs = fmaf(qq + pu->data[h * hd + e], k[(int64_t)j * d + h * hd + e], cs);     /* :212 */
                    ps = fmaf(qq + pv->data[h * hd + e], PT[((int64_t)h * hd + e) * P + off + j], ps);   /* :214-224 */
                }
                float sc = (cs + ps) * scale;                                                       /* :226 */
                const int dist = abs_pos - j;
                if ((s->att_left >= 0 || s->att_right >= 0) && (dist > s->att_left || -dist > s->att_right)) sc = -1e9f;   /* :231-247 */
                row[j] = sc;
                mx = sc > mx ? sc : mx;
            }
            for (int j = 0; j < kv; ++j) row[j] = orc_expf(row[j] - mx);
            const float sum = orc_sum64(row, kv, 1);
            for (int j = 0; j < kv; ++j) row[j] = row[j] / sum;
            for (int e = 0; e < hd; ++e) {
                float acc = 0.0f;
                for (int j = 0; j < kv; ++j) acc = fmaf(row[j], v[(int64_t)j * d + h * hd + e], acc);   /* :250 */
                ctx[(int64_t)i * d + h * hd + e] = acc;
            }
        }
    free(row);
    linear_t(cf->gemm_bf16, wo, bo, c, ctx, d, y, d, 0);                                         /* :255 */
    for (int64_t i = 0; i < (int64_t)c * d; ++i) x[i] = x[i] + y[i];
    free(n); free(q); free(k); free(v); free(ctx); free(y);
    return 0;
}

/* CausalConformerConvModule::forward_cached -- src/streaming_encoder.cpp:41-78.  x[c][d] in place. */
static int stream_conv(orc_stream *s, int layer, float *x, int c) {
    orc_model *m = s->m;
    const orc_config *cf = &m->cfg;
    const int d = cf->d_model, Kc = cf->conv_k, cl = Kc - 1;
    orc_tensor *ng = getf(m, "encoder_.layers_.%d.conv_.norm_.weight", layer), *nb = getf(m, "encoder_.layers_.%d.conv_.norm_.bias", layer);
    orc_tensor *w1 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv1_.weight", layer), *b1 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv1_.bias", layer);
    orc_tensor *wd = getf(m, "encoder_.layers_.%d.conv_.depthwise_conv_.weight", layer), *bd = getf(m, "encoder_.layers_.%d.conv_.depthwise_conv_.bias", layer);
    orc_tensor *bng = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.weight", layer), *bnb = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.bias", layer);
    orc_tensor *bnm = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.running_mean", layer), *bnv = getf(m, "encoder_.layers_.%d.conv_.batch_norm_.running_var", layer);
    orc_tensor *w2 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv2_.weight", layer), *b2 = getf(m, "encoder_.layers_.%d.conv_.pointwise_conv2_.bias", layer);
    if (!ng || !nb || !w1 || !b1 || !wd || !bd || !bng || !bnb || !bnm || !bnv || !w2 || !b2) return -1;
    float *n = (float *)xmalloc((size_t)c * d * sizeof(float)), *g2 = (float *)xmalloc((size_t)c * 2 * d * sizeof(float));
    float *cat = (float *)xmalloc((size_t)(cl + c) * d * sizeof(float)), *dw = (float *)xmalloc((size_t)c * d * sizeof(float));
    float *y = (float *)xmalloc((size_t)c * d * sizeof(float));
    layer_norm(x, c, d, ng->data, nb->data, cf->ln_eps, n);
    linear_t(cf->gemm_bf16, w1, b1, c, n, d, g2, 2 * d, 0);
    if (s->has_conv[layer]) memcpy(cat, s->cc[layer], (size_t)cl * d * sizeof(float));   /* prepend the cache :51-63 */
    else memset(cat, 0, (size_t)cl * d * sizeof(float));
    for (int t = 0; t < c; ++t)
        for (int ch = 0; ch < d; ++ch) cat[(int64_t)(cl + t) * d + ch] = g2[(int64_t)t * 2 * d + ch] * orc_sigmoidf(g2[(int64_t)t * 2 * d + d + ch]);   /* glu :49 */
    memcpy(s->cc[layer], cat + (int64_t)c * d, (size_t)cl * d * sizeof(float));            /* last Kc-1 frames :66-69 */
    s->has_conv[layer] = 1;
    for (int t = 0; t < c; ++t)
        for (int ch = 0; ch < d; ++ch) {
            float acc = 0.0f;                                                                /* depthwise, no padding :71 */
            for (int kk = 0; kk < Kc; ++kk) acc = fmaf(wd->data[ch * Kc + kk], cat[(int64_t)(t + kk) * d + ch], acc);
            float v = acc + bd->data[ch];
            const float rstd = 1.0f / sqrtf(bnv->data[ch] + cf->bn_eps);
            v = fmaf((v - bnm->data[ch]) * rstd, bng->data[ch], bnb->data[ch]);
            dw[(int64_t)t * d + ch] = orc_siluf(v);
        }
    linear_t(cf->gemm_bf16, w2, b2, c, dw, d, y, d, 0);
    for (int64_t i = 0; i < (int64_t)c * d; ++i) x[i] = x[i] + y[i];
    free(n); free(g2); free(cat); free(dw); free(y);
    return 0;
}

/* StreamingFastConformerEncoder::forward_chunk -- src/streaming_encoder.cpp:430-472 (+ CausalConvSubsampling::forward_cached
 * :348-385, StreamingConformerBlock::forward_cached :289-301).  mel[n_frames][n_mels] -> enc[c][d]; returns c (0: cached).
 * xscaling (:444-447; on only in the Sortformer NEST config) follows orc_model_set_encoder. */
int orc_stream_encode(orc_stream *s, const float *mel, int n_frames, float *enc, int max_out) {
    orc_model *m = s->m;
    const orc_config *cf = &m->cfg;
    const int F = cf->mel_bins, d = cf->d_model;
    const int total = s->n_mel_cache + n_frames;
    float *all = (float *)xmalloc((size_t)(total > 0 ? total : 1) * F * sizeof(float));
    memcpy(all, s->mel_cache, (size_t)s->n_mel_cache * F * sizeof(float));
    memcpy(all + (int64_t)s->n_mel_cache * F, mel, (size_t)n_frames * F * sizeof(float));
    const int consumable = (total / 8) * 8;                                          /* :365 */
    const int left = total - consumable;
    memcpy(s->mel_cache, all + (int64_t)consumable * F, (size_t)left * F * sizeof(float));
    s->n_mel_cache = left;
    if (consumable == 0) { free(all); return 0; }
    const int c = orc_subsampled_len(consumable);
    if (c > max_out) { free(all); return orc_fail("orc_stream_encode: %d frames > max_out %d", c, max_out); }
    if (orc_subsampling(m, all, 1, consumable, enc, NULL, NULL) < 0) { free(all); return -1; }
    free(all);
    if (m->xscaling) {
        const float scale = sqrtf((float)d);
        for (int64_t i = 0; i < (int64_t)c * d; ++i) enc[i] = enc[i] * scale;
    }
    const int Tp = s->att_left + c, P = 2 * Tp - 1;                                  /* :452-454 */
    float *pe = (float *)xmalloc((size_t)P * d * sizeof(float));
    orc_pos_emb(Tp, d, pe);
    for (int l = 0; l < cf->n_layers; ++l) {
        if (prepare_layer(m, l)) { free(pe); return -1; }
        float *PT = pos_proj_heads(m, l, Tp, pe, 0);                                 /* streaming: fp32 attention arithmetic in either mode */
        if (!PT) { free(pe); return -1; }
        int r = feed_forward(m, l, "ffn1_", enc, c);
        if (!r) r = stream_attention(s, l, enc, c, PT, P);
        if (!r) r = stream_conv(s, l, enc, c);
        if (!r) r = feed_forward(m, l, "ffn2_", enc, c);
        free(PT);
        if (r) { free(pe); return -1; }
        orc_tensor *fg = getf(m, "encoder_.layers_.%d.final_norm_.weight", l), *fbb = getf(m, "encoder_.layers_.%d.final_norm_.bias", l);
        if (!fg || !fbb) { free(pe); return -1; }
        layer_norm(enc, c, d, fg->data, fbb->data, cf->ln_eps, enc);
    }
    free(pe);
    return c;
}

/* rnnt_streaming_decode_chunk -- src/eou.cpp:17-98: the TDT greedy loop of tdt.cpp on one encoder chunk with the LSTM state
 * and the last token carried across chunks; frames are reported relative to the stream (frame_offset), end frames are not
 * clamped to the chunk; a duration that skips past the end of the chunk is simply lost (:31-33,95). */
/* step_margin / step_label [step_cap] (optional): top-1 / top-2 margin and chosen label of EVERY decision of the chunk, in order (the decisions
 * of the tolerance-class mode's token statement, oracle/tolerance.py); *n_steps = how many there were. */
int orc_stream_decode_ex(orc_stream *s, const float *enc, int c, int max_tokens, int32_t *ids, int32_t *start, int32_t *end, float *conf,
                         float *step_margin, int32_t *step_label, int step_cap, int32_t *n_steps) {
    int32_t len = 0, steps = 0;
    const int cap = c * (s->m->cfg.max_symbols + 1) + 16;
    float mm = 0.0f;                                                /* (the per-decision records are kept where the minimum is asked for) */
    const int r = tdt_greedy_ex(s->m, enc, 1, c, max_tokens, cap, ids, &len, start, end, conf, &steps, NULL, s->hc, &s->token, 0, NULL, 0.0f,
                                (step_margin || step_label) ? &mm : NULL, step_margin, step_label, step_cap);
    if (r || len < 0) return orc_fail("orc_stream_decode: decode cap hit");
    for (int i = 0; i < len; ++i) { if (start) start[i] += s->frame_offset; if (end) end[i] += s->frame_offset; }
    s->frame_offset += c;
    if (n_steps) *n_steps = steps;
    return len;
}
int orc_stream_decode(orc_stream *s, const float *enc, int c, int max_tokens, int32_t *ids, int32_t *start, int32_t *end, float *conf) {
    return orc_stream_decode_ex(s, enc, c, max_tokens, ids, start, end, conf, NULL, NULL, 0, NULL);
}
/* The same chunk loop (src/eou.cpp:17-98) along a GIVEN decision path -- orc_tdt_score with the stream's carried LSTM state and last token
 * (StreamingDecodeState, eou.hpp:80-87): the decisions of the chunk's steps are labels_in[k] / dur_in[k] (NULL: the greedy ones, reported in
 * labels_out / dur_out), the joint's log-softmax outputs of every step are recorded, and the state the stream carries into the next chunk is
 * the one the reference's loop holds after deciding that way.  Returns the number of steps walked (the chunk is over when the frame pointer
 * leaves it -- a duration that skips past its end is lost, :31-33,95 -- or after n_steps). */
int orc_stream_score(orc_stream *s, const float *enc, int c, const int32_t *labels_in, const int32_t *dur_in, int n_steps, int32_t *labels_out,
                     int32_t *dur_out, float *label_lp, float *dur_lp) {
    const int k = tdt_score_ex(s->m, enc, c, labels_in, dur_in, n_steps, labels_out, dur_out, label_lp, dur_lp, s->hc, &s->token);
    if (k < 0) return -1;
    s->frame_offset += c;
    return k;
}

/* Sortformer::diarize_chunk (src/sortformer.cpp:123-150): forward_chunk of the NEST encoder with the stream's caches, then
 * projection / transformer / head on THIS chunk's frames only.  feats[n_frames][mel] -> probs[c][S]; returns c (0: buffered). */
int orc_sortformer_chunk(orc_stream *s, const float *feats, int n_frames, int n_tlayers, int n_theads, int pre_ln, int has_final_norm,
                         float *probs, int max_out) {
    const int d = s->m->cfg.d_model;
    float *enc = (float *)xmalloc((size_t)(max_out > 0 ? max_out : 1) * d * sizeof(float));
    const int c = orc_stream_encode(s, feats, n_frames, enc, max_out);
    if (c <= 0) { free(enc); return c; }
    const int r = sortformer_head(s->m, enc, 1, c, n_tlayers, n_theads, pre_ln, has_final_norm, probs);
    free(enc);
    return r == 0 ? c : -1;
}
