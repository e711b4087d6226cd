// parakeet.cpp_amd/csrc/kernels/kernels.hpp -- host-side launchers of the gfx950 kernels.
// Every launcher enqueues on `s` and returns; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace pk {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting and the launchers run on one host thread per device in a
// pk_group: the largest size set so far is kept per (launcher, device).  `slots` is the launcher's own static array.
constexpr int kMaxDevices = 16;
struct DynLdsSlots { std::atomic<size_t> set[kMaxDevices]; };
inline void ensure_dyn_lds(DynLdsSlots &slots, const void *kernel, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) {          // no slot for this device: set the attribute every time, never alias another device's slot
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        return;
    }
    std::atomic<size_t> &cur = slots.set[dev];
    if (bytes > cur.load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        cur.store(bytes, std::memory_order_release);
    }
}

// ---- ragged (mixed-length, packed) batches -------------------------------------------------------------------------------------
// The reference's roadmap "batch inference: pad + length-mask" (README.md:513) done WITHOUT padding: utterance b of a packed tensor owns
// rows [off[b], off[b] + n[b]) -- GEMMs, LayerNorm and their epilogues are row-wise and never notice; kernels that look across rows (mel
// framing, the 3x3 / depthwise convolutions, attention, the greedy decoders) take the per-utterance extents below, so every utterance sees
// exactly the zero padding, position table and softmax extent of a single-clip run: bit-identical per clip.  A null pointer in the first
// member = uniform batch (the launchers' plain B / T arguments apply).
struct RagUnit { int b, r0; };                 // one strip of rows of ONE utterance: utterance b, first local row r0
struct RagUnits { const RagUnit *u = nullptr; int count = 0; };
struct MelRag { const int64_t *pcm_off = nullptr; const int *Tm = nullptr, *Tm_off = nullptr; int max_frames = 0;   // [B+1] samples, [B] mel frames, [B+1]
                const int *Tm_pad_off = nullptr; };   // [B+1] prefix sums of mel_logmel_pitch(Tm[b]): where clip b's log-mel block starts (in frames)
struct SubRag {                                 // conv subsampling: mel frames -> H2 = rows after conv1 / dw1 -> T rows after dw2
    RagUnits strips;                            // conv1_dw1: strips of strip_rows rows of H2 ; dw2: one unit per output row
    int strip_rows = 0;
    const int *Tm = nullptr, *Tm_off = nullptr, *H2 = nullptr, *H2_off = nullptr, *T = nullptr, *T_off = nullptr;
};
struct SeqRag {                                 // encoder frames: attention row blocks (32 rows) / depthwise-conv strips / decoders
    RagUnits units;
    const int *T = nullptr, *T_off = nullptr;  // [B], [B+1]
    int T_max = 0;                              // longest utterance of the batch (LDS sizing)
    int pos_T = 0;                              // the position table was built for pos_T >= T_max frames: utterance b reads it from row pos_T - T[b]
};

// ---- mel front end (src/audio.cpp:100-158) ----------------------------------------------------
constexpr int kMelMaxTaps = 768;      // packed filterbank taps staged in LDS by the mel kernel (sum of the band widths; 80 / 128 bins: ~590)
struct MelTables {
    const float *window;   // [512] symmetric Hann(400) zero-padded to n_fft, placed per switch A1 (pk_config.stft_window_centered)
    const float *window_left;  // [512] the same window left-aligned: the streaming preprocessor's frames (center=false)
    const float *tw_re;    // [256] cos(2 pi k / 512)
    const float *tw_im;    // [256] -sin(2 pi k / 512)
    const float *fb;       // [257][n_mels] Slaney filterbank (src/audio.cpp:40-94)
    const int *f_lo;       // [n_mels] first / last non-zero fft bin of each filter
    const int *f_hi;
    const float *fbc;      // the non-zero band of every filter, packed: fbc[fb_off[m] + f - f_lo[m]] = fb[f][m]
    const int *fb_off;     // [n_mels + 1]
    int fb_nnz;            // fb_off[n_mels] (<= kMelMaxTaps)
    int n_mels;
    int power_via_abs;     // switch A2
};
// rag set: clip b = pcm[rag.pcm_off[b] .. rag.pcm_off[b+1]), its log-mel [n_mels][pitch of rag.Tm[b]] at logmel + n_mels * rag.Tm_pad_off[b] (n_samples / n_frames ignored)
// The log-mel intermediate is [clip][n_mels][pitch], pitch = mel_logmel_pitch(n_frames) = n_frames rounded up to 16: every run of 16 frames a
// workgroup stores is then one aligned 64-byte segment (the un-padded rows -- 1001 floats for 10 s -- put every run across sector boundaries:
// 1.44x write traffic).  Only mel_normalize (and the debug read-outs of pk_mel) read it.
__host__ __device__ inline int mel_logmel_pitch(int n_frames) { return (n_frames + 15) & ~15; }
void launch_mel_logmel(const float *pcm, int B, int64_t n_samples, int n_frames, const MelTables &t, float *logmel, hipStream_t s, const MelRag &rag = MelRag());
// StreamingAudioPreprocessor::process_chunk (src/audio.cpp:222-252) on pre-emphasised buffers pre[B][n_samples]:
// n_frames = (n_samples - 400) / 160 + 1 frames -> log-mel [B][n_frames][n_mels] (no normalisation)
void launch_mel_stream(const float *pre, int B, int64_t n_samples, int n_frames, const MelTables &t, float *logmel_tf, hipStream_t s);
void launch_mel_normalize(const float *logmel, int B, int n_mels, int n_frames, int normalize, float *feats, hipStream_t s, const MelRag &rag = MelRag());

// ---- fp32 MFMA GEMM: out = epi(A[M][K] * W[N][K]^T + bias), natural-k fma chains ----------------
enum GemmEpi { EPI_NONE = 0, EPI_RELU = 1, EPI_SILU = 2, EPI_RESID = 3, EPI_GLU = 4 };
// Streaming conv module: the causal depthwise conv (kernel 9) + BatchNorm + SiLU of
// CausalConformerConvModule::forward_cached (src/streaming_encoder.cpp:41-78) run in the GLU epilogue of pw1 (tolerance-class mode:
// gemm_smallm_bf16.hip; exact mode: gemm_smallm_ln_kernel of gemm_smallm.hip): the lane that
// finishes column ch of a stream's c new frames holds everything the conv of (stream, ch) needs next to the stream's cached 8 frames -- no
// exchange, one launch less per block.  Rows of the product = [S][c] stream-major; GemmArgs::out receives the conv module's activations
// (the GLU values themselves are not stored).  Same operations in the same order as stream_dwconv_kernel: bit-identical.
struct DwTail {
    const float *cache_in = nullptr; float *cache_out = nullptr;     // [S][8][d]: the last 8 GLU rows of every stream before / after this chunk
    int has_cache = 0, c = 0;                                        // first chunk: zero left padding; c = new frames per stream (1, 2 or 4)
    const float *w = nullptr /* [9][d] */, *bias = nullptr, *bn_mean = nullptr, *bn_rstd = nullptr, *bn_g = nullptr, *bn_b = nullptr;
    int out_sigma = 0;                                               // the activations' columns in the sigma layout (exact mode: pw2 reads a_sigma rows)
};

struct GemmArgs {
    const float *A; int64_t lda;
    const float *W; int64_t ldw;
    const float *bias;          // [N] (GLU: [2N]) or nullptr
    float *out; int64_t ldo;
    const float *resid; int64_t ldr; float alpha;   // EPI_RESID: out = resid + alpha * (acc + bias)
    int M, N, K;                // N = output width (GLU: W has 2N rows, a-part rows [0,N), gate rows [N,2N))
    // optional output re-mapping (0 = off): offset = (row / remap_rows) * remap_gs + (row % remap_rows) * remap_rs + col * remap_cs
    // (used to write the last subsampling conv straight into the permute(0,2,1,3)+reshape layout, src/encoder.cpp:235-238)
    int remap_rows = 0; int64_t remap_gs = 0, remap_rs = 0, remap_cs = 0;
    // output columns < sigma_cols (a multiple of 16) are written in the "sigma" layout: inside every block of 16 columns the
    // 4x4 index matrix is transposed (col -> (col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)).  The attention kernel reads
    // Q, K and the projected position table that way: a lane's float4 then holds its operands of 4 consecutive MFMA steps.
    int sigma_cols = 0;
    // bf16 mode (launch_gemm_bf16 / gemm_bf16.hpp) only: A points to bf16 activations [M][lda] the producer already rounded (RNE, the
    // rounding the staging path would apply -- same operand values, half the bytes); out points to a bf16 buffer [M][ldo] (the next
    // product's A operand; row-major wide outputs only).
    int a_bf16 = 0, out_bf16 = 0;
    // bf16 mode, direct-to-LDS kernels (gemm_bf16_glds.hpp) only: the bf16 activation tensor between a producer's register epilogue and the
    // next product's LDS-DMA lives in BLOCKS of 32 rows x 16 columns (1 KB, row-major inside; blocks ordered [row / 32][col / 16]): one store
    // instruction of the producer's epilogue (32 rows x 2 x 16 bytes) then writes ONE contiguous KB instead of 32 row slivers of 32 bytes,
    // and the consumer's DMA (which gathers 16-byte chunks by per-lane address anyway) reads it in 256-byte runs.  out_blocked: `out` is
    // written that way (ldo = the row length in elements, a multiple of 16; rows padded to a multiple of 32 by the allocation);
    // a_blocked: `A` is read that way (lda = its row length).  Set by the engine for fc1 -> fc2 when gemm_bf16_blocked_handoff() says both take
    // those kernels.
    int out_blocked = 0, a_blocked = 0;
    // bf16 mode, direct-to-LDS kernels, EPI_RESID on one tile per workgroup: the product is accumulated ONTO the residual -- the accumulators start
    // from resid / alpha + bias (loaded beside the first K tiles) and the epilogue stores alpha * acc: the residual's bytes leave HBM while the K
    // loop runs instead of at the tail next to the output stores.  Same value in real arithmetic; rounded differently (tolerance-class mode only).
    // EXPERIMENT of round 5, measured slower (gemm.hip: bf16_glds_flags bit 8): never set by the production library.
    int resid_init = 0;
    // small-M bf16 kernel only (gemm_smallm_bf16.hip): the bf16 activation rows between two of its products (fc1 -> fc2 of a streaming chunk) in
    // OPERAND TILES of 8 rows: per (8 rows, 32 k) 512 bytes [lane32 = (row & 7) + 8 * (k / 8 & 3)][8 bf16], tiles ordered [rows / 8][K / 32] -- the
    // consumer's load instruction reads whole lines (512 B, or two runs of 512 B for 16 rows) instead of 8 / 16 row segments of 64 bytes.
    // M % 8 == 0, row length % 32 == 0.  out_t8: `out` is written that way (ldo = the row length), a_t8: `A` is read that way (lda = the row length).
    int out_t8 = 0, a_t8 = 0;
    // bf16 mode only: SiLU / sigmoid of the epilogue on the hardware exp2 / rcp (1 ulp) instead of the fixed polynomial + IEEE division of the
    // numerics contract -- that mode is compared with the oracle within a bf16-epsilon-class tolerance, not bit for bit, and at bf16 MFMA rates
    // the 38-operation SiLU is as expensive as the product itself (fc1 of tdt-600m: ~48 us of VALU against 40 us of MFMA).
    int fast_act = 0;
    // small-M kernel (gemm_smallm.hip, M <= kSmallMRows) only: W_sig = a copy of W (launch_sigma_copy; N % 16 == 0, K % 64 == 0) tiled in the kernel's load
    // order -- per (16-row tile, 64-k chunk) one 4 KB block [q][lane][4] with the K axis in the sigma layout; a_sigma = A's K axis is in the sigma
    // layout (its producer wrote it that way: sigma_cols, LayerNorm mode 2, the streaming attention / conv kernels).  Both set: a lane's 16-byte
    // load IS its operand of four consecutive MFMA steps -- no transposes on the chain -- and every weight load is 1 KB of consecutive addresses.
    const float *W_sig = nullptr; int a_sigma = 0;
    // small-M bf16 kernel (gemm_smallm_bf16.hip) only: a copy of the bf16 weights in the kernel's OPERAND TILES (launch_tile_copy_bf16: per
    // (16 rows, 32 k) one KB in lane order, tiles [rows / 16][K / 32]; rows % 16 == 0, K % 32 == 0) -- every weight load instruction reads one
    // contiguous KB.  W stays set (the natural layout: every other kernel reads that).
    const float *W_t16 = nullptr;
    // small-M kernels only (gemm_smallm_bf16.hip; gemm_smallm.hip with W_sig): A = the UN-normalised fp32 rows, the LayerNorm (gamma ln_g[K], beta
    // ln_b[K], ln_eps) of the product's input is applied while the rows are staged -- out = epi(LN(A) W^T + bias) (bf16 mode: bf16(LN(A)) W16^T).
    // Callers check gemm_smallm_bf16_ln_applies() / gemm_smallm_ln_applies().
    const float *ln_g = nullptr, *ln_b = nullptr; float ln_eps = 0.0f;
    // Tile kernel of the fp32 path (gemm_pipe.hpp, LNA; round 6): with ln_g / ln_b set, ln_stats = {mean, rstd} of every row of A (launch_layernorm_stats) and
    // the normalisation y = fma((x - mean) * rstd, gamma, beta) is applied while the A tile is staged -- bit for bit the rows launch_layernorm would have
    // written.  Callers check gemm_ln_stats_applies().
    const float *ln_stats = nullptr;
    // small-M kernels with ln_g set (gemm_smallm_bf16.hip; exact mode: gemm_smallm_ln_kernel, bit for bit the separate launch): ANOTHER LayerNorm in front of the folded one -- out = epi(bf16(LN(LN(A; pre_g, pre_b); ln_g, ln_b)) W16^T + bias):
    // a block's final_norm_ folded into the first product of the next block (streaming, tolerance-class mode: one launch less per block).  pre_out
    // (optional, fp32 [M][pre_ldo], NOT the buffer A lives in: other workgroups still read A) receives LN(A; pre_g, pre_b) -- the residual stream
    // of the block that starts here -- written by the first few column tiles, one k-step (chunk) of every row each.
    const float *pre_g = nullptr, *pre_b = nullptr; float *pre_out = nullptr; int64_t pre_ldo = 0;
    // small-M kernels with the LayerNorm folded in (exact mode) / small-M bf16 kernel, EPI_GLU only: HOST pointer to the depthwise-conv tail of
    // the epilogue (read during the launch call).  Callers check gemm_smallm_dw_applies() / gemm_smallm_bf16_dw_applies().
    const DwTail *dw_tail = nullptr;
};
constexpr int kSmallMRows = 1536;  // launch_gemm: products with M <= this (and K % 64 == 0) run on gemm_smallm.hip.  Measured with the two-row-tile
                                   // variant in place (110m encoder; reference protocol, batch 1: M = 626 5.6 ms on it vs 6.7 on the tile kernels,
                                   // M = 751 6.1 vs 6.9, M = 1251 12.4 vs 15.1; pipelined batches of 8 / 12 / 16 clips, M = 1008 / 1512 / 2016:
                                   // 6.75 vs 6.50, 9.14 vs 10.67, 11.35 vs 10.93 ms per step)
// src [rows][ld] -> dst rows x K floats in the W_sig tiling (rows % 16 == 0, K % 64 == 0)
void launch_sigma_copy(const float *src, float *dst, int64_t rows, int K, int64_t ld, hipStream_t s);
// bf16 src [rows][ld] -> dst rows x K bf16 in the W_t16 operand tiles (rows % 16 == 0, K % 32 == 0, ld % 8 == 0)
void launch_tile_copy_bf16(const float *src16, float *dst16, int64_t rows, int K, int64_t ld, hipStream_t s);
void launch_gemm(const GemmArgs &a, int epi, hipStream_t s);
// fp32 tile kernel with the LayerNorm of its input rows applied from per-row statistics (GemmArgs::ln_stats): large batches (M > kSmallMRows), K = the row
// length (a multiple of 32), wide outputs (the 128 x 128 single-buffered tile: fc1, qkv, pw1 + GLU), epilogues none / ReLU / SiLU / GLU
bool gemm_ln_stats_applies(const GemmArgs &a, int epi);
// bf16 mode: true when the product (M x N x K, bf16 A already in HBM, epilogue `epi`) runs on a direct-to-LDS kernel that can write
// (producer = true: register epilogue) / read (producer = false: LDS-DMA) the blocked activation layout of GemmArgs::out_blocked / a_blocked
bool gemm_bf16_blocked_handoff(int M, int N, int K, int epi, bool producer);
// fp32 small-M chain kernel with GemmArgs::ln_g set (kernels/gemm_smallm.hip: gemm_smallm_ln_kernel): A = the un-normalised rows, K = 512 / 1024 = the
// row length, tiled weights W_sig, epi none / relu / silu / glu -- bit for bit LayerNorm + product.  Callers check this first.
bool gemm_smallm_ln_applies(const GemmArgs &a, int epi);
bool gemm_smallm_pre_applies(const GemmArgs &a, int epi);                      // ... with GemmArgs::pre_g (a second norm in front, bit for bit the separate LayerNorm launch)
bool gemm_smallm_dw_applies(const GemmArgs &a, int epi, int c, int kc);        // ... with GemmArgs::dw_tail (GLU, conv kernel 9, c = 1 / 2 / 4 frames per stream)
// same contract with bf16 operands and fp32 accumulation: a.W points to bf16 weights [N][K] (rounded once at upload), A is
// rounded to bf16 while it is staged; K % 64 == 0.  Not bit-identical to the fp32 chain (kernels/gemm_bf16.hpp).
void launch_gemm_bf16(const GemmArgs &a, int epi, hipStream_t s);
// the same for a handful of rows (streaming chunks in the tolerance-class mode): kernels/gemm_smallm_bf16.hip -- one workgroup per 16 output
// columns, K split over its waves, every weight byte requested up front.  launch_gemm_bf16 routes the shapes gemm_smallm_bf16_applies() accepts
// (M <= kSmallMRowsBf16, K % 256 == 0, row-major output) there.
constexpr int kSmallMRowsBf16 = 128;
bool gemm_smallm_bf16_applies(const GemmArgs &a, int epi);
bool gemm_smallm_bf16_dw_applies(const GemmArgs &a, int epi, int c, int kc);   // ... with GemmArgs::dw_tail: GLU, conv kernel 9, c = 1 / 2 / 4 frames per stream
bool gemm_smallm_bf16_ln_applies(const GemmArgs &a, int epi);
bool gemm_smallm_bf16_pre_applies(const GemmArgs &a, int epi);    // ... and GemmArgs::pre_g (a second norm in front): SiLU products, slices of 256 k     // ... with GemmArgs::ln_g set: K = 256 * (1 .. 8 waves; GLU: 4), fp32 rows
void launch_gemm_smallm_bf16(const GemmArgs &a, int epi, hipStream_t s);
double gemm_flops(const GemmArgs &a, int epi);

// ---- conv subsampling (src/encoder.cpp:219-241), channels-last ---------------------------------------
// rag set (strips = units of sub_conv1_dw1_strip_rows(...) rows of H2): packed feats [sum Tm][F] -> out [sum H2][W2][C]; B / Tm ignored
int sub_conv1_dw1_strip_rows(int64_t total_h2_rows);       // output rows per strip the launcher will use for a batch with that many H2 rows
void launch_sub_conv1_dw1(const float *feats, int B, int Tm, int F, int C, const float *w1, const float *b1, const float *wd,
                          const float *bd, float *out, hipStream_t s, const SubRag &rag = SubRag());
// rag set (strips = one unit per output row): packed in [sum H2][W][C] -> out [sum T][Wo][C]; B / H ignored
void launch_sub_dw(const float *in, int B, int H, int W, int C, const float *wd, const float *bd, float *out, hipStream_t s, const SubRag &rag = SubRag());

// ---- conformer pieces ---------------------------------------------------------------------------------
// qkv[B*T][3d]: q and k thirds in the sigma column layout.  pos == nullptr: plain multi-head attention (src/transformer.cpp:38),
// no position term, bias_u / bias_v ignored.  scale <= 0: 1/sqrt(d / n_heads).
size_t relpos_attention_lds_bytes(int T, int hd);     // 0: unsupported head size
int relpos_attention_max_frames(int hd);              // longest sequence one workgroup's LDS score block can hold
size_t relpos_attention_scratch_bytes(int B, int T, int n_heads, int hd);   // long sequences: score blocks in global scratch (hd 64 / 128)
// pos_row0: first row of the table to use (the table may have been built for a longer sequence: row p of a T-frame table is row
// p + T_table - T of the longer one, engine.cpp ensure_pos_tables).  rag set (units = 32-row blocks): qkv / ctx are packed [sum T] rows,
// utterance b attends over its own T[b] frames with table rows from rag.pos_T - T[b]; B / T / pos_row0 ignored, LDS sized by rag.T_max.
void launch_relpos_attention(const float *qkv, int B, int T, int d, int n_heads, const float *pos, const float *bias_u,
                             const float *bias_v, float *ctx, hipStream_t s, float scale = 0.0f, float *scratch = nullptr, int ctx_bf16 = 0,
                             int pos_row0 = 0, const SeqRag &rag = SeqRag());
size_t relpos_attention_scratch_bytes_units(int64_t n_units, int T_max, int n_heads, int hd);   // ragged form of relpos_attention_scratch_bytes
// The tolerance-class (pk_config.gemm_bf16) attention, kernels/attention_bf16.hip: q / k / v as bf16 [B*T][3d] (natural columns, written by the
// qkv GEMM), the projected position table of the layer as bf16 [2T-1][d], cvec[h][p] = (v_h - u_h) . P_p (fp32, launch_pos_cvec), ctx as bf16.
// Head sizes 64 and 128; relpos_attention_bf16_lds_bytes returns 0 for any other.
size_t relpos_attention_bf16_lds_bytes(int T, int hd);
void launch_pos_cvec(const void *pos_bf16, const float *bias_u, const float *bias_v, int P, int d, int n_heads, float *cvec, hipStream_t s);
// pos_T: sequence length the table / cvec were built for (>= T; 0 = T).  rag set (units = the kernel's query-row blocks,
// relpos_attention_bf16_block_rows()): packed rows, per-utterance T.
int relpos_attention_bf16_block_rows(int hd);
void launch_relpos_attention_bf16(const void *qkv_bf16, int B, int T, int d, int n_heads, const void *pos_bf16, const float *cvec, const float *bias_u,
                                  void *ctx_bf16, hipStream_t s, int pos_T = 0, const SeqRag &rag = SeqRag());
void launch_dwconv_bn_silu(const float *g, int B, int T, int d, int kc, const float *w, const float *bias, const float *bn_mean,
                           const float *bn_rstd, const float *bn_g, const float *bn_b, float *out, hipStream_t s, int out_bf16 = 0,
                           const SeqRag &rag = SeqRag());
int dwconv_strip_frames(int64_t total_rows);   // frames per strip launch_dwconv_bn_silu uses for a batch of that many rows (rag.units granularity)

// ---- streaming encoder pieces (src/streaming_encoder.cpp) -----------------------------------------------------------------------
// StreamingConformerAttention::forward_cached (:162-272) core for S streams x c query rows: keys / values = nc cached rows
// followed by the c rows of this chunk (qkv_new[S*c][3d], natural columns); position scores are the rightmost kv = nc + c
// columns of (q+v) P^T WITHOUT rel_shift (:215-224), masked to the [left, right] context (:226-247).  ctx[S*c][d].
void launch_stream_attention(const float *qkv_new, const float *kcache, const float *vcache, int cache_rows, int S, int c, int nc, int d,
                             int n_heads, const float *pos /*[P][d]*/, int P, const float *bias_u, const float *bias_v, int att_left,
                             int att_right, float *ctx, hipStream_t s, float *cache_k_out = nullptr, float *cache_v_out = nullptr, int keep_max = 0,
                             int ctx_sigma = 0 /* ctx columns in the sigma layout (GemmArgs::a_sigma of the out-projection) */);
// (cache_k_out / cache_v_out set: the same launch also rotates the K / V caches of every (stream, head) into those buffers -- the last
//  min(keep_max, nc + c) rows of [cache ; new], what two launch_stream_cache_update calls would write)
// new cache = the last min(keep_max, nc + c) rows of [cache(nc rows) ; new(c rows)]  (:193-209); row stride of both caches: cache_rows*d
void launch_stream_cache_update(const float *cache_in, int nc, const float *qkv_new, int col0, int S, int c, int d, int cache_rows,
                                int keep_max, float *cache_out, hipStream_t s);
// CausalConformerConvModule::forward_cached middle (:51-73): depthwise conv over [cache(K-1 rows, zeros when !has_cache) ; g(c rows)],
// BatchNorm, SiLU -> out[S*c][d]; cache_out = last K-1 rows of the concatenation.
void launch_stream_dwconv(const float *g, const float *cache_in, int has_cache, int S, int c, int d, int kc, const float *w, const float *bias,
                          const float *bn_mean, const float *bn_rstd, const float *bn_g, const float *bn_b, float *out, float *cache_out,
                          hipStream_t s, int out_sigma = 0 /* out channels in the sigma layout */);

// ---- decoders -------------------------------------------------------------------------------------------
void launch_logsoftmax_argmax(const float *logits, int64_t rows, int ld, int n, float *lp_out, int *best_idx, float *best_lp, hipStream_t s);
// outputs [B][pitch] (pitch 0 = T).  rag set: best_idx / best_lp are packed [sum T], utterance b has rag.T[b] frames from row rag.T_off[b].
void launch_ctc_collapse(const int *best_idx, const float *best_lp, int B, int T, int blank, int *ids, int *lens, int *start, int *end,
                         float *conf, hipStream_t s, int pitch = 0, const SeqRag &rag = SeqRag());
// ContextTrie of the phrase boosting (reference src/phrase_boost.cpp:9-66) in CSR form: the children of node i are the entries
// [off[i], off[i+1]) of (tok, node).  Every utterance carries its active-state set (the root plus at most one node per trie
// depth, so <= kTrieMaxActive for phrases of < kTrieMaxActive tokens).  off == nullptr: boosting is off.
constexpr int kTrieMaxActive = 64;
struct TrieDev {
    const int *off, *tok, *node;
    int n_nodes;
    float boost;
    int *act, *n_act;               // [B][kTrieMaxActive], [B]
};
void launch_ctc_boosted(const float *logp, int B, int T, int V, int blank, const TrieDev &trie, int *ids, int *lens, int *start, int *end,
                        float *conf, hipStream_t s, int pitch = 0, const SeqRag &rag = SeqRag());
struct TdtState {
    int B, T, V, D, L, Hp, blank, max_symbols, max_tokens, max_steps;
    TrieDev trie;
    int keep_state;                 // streaming chunks (src/eou.cpp:17-98): the last token and h / c are carried in, end frames are not clamped
    int durations[8];
    const float *logits;            // [B][V+D]
    float *h, *c;                   // committed LSTM state [L][B][Hp]
    const float *hn, *cn;           // candidates of this step
    int *token, *t, *nsym, *n_out, *steps, *done, *lens, *done_count;
    int *ids, *start, *end;         // [B][max_tokens]
    float *conf;
    int h_bf16;                     // tolerance-class mode: h / hn are bf16 arrays (decode_gemv_bf16.hip); c / cn stay fp32
    float *margin;                  // [B] or null: running min over the utterance's decisions of (top-1 - top-2) label log-prob
    // Prediction-net caching (null = off).  After a BLANK the reference reverts the LSTM state and keeps the token (src/tdt.cpp:69-93), so the next
    // step's prediction.step and pred_proj are functions of unchanged inputs: bit for bit the values of the step before.  need[b] = 1 says
    // utterance b's next step must run them (set at the start and after every emitted token); the cell / pred_proj launches compact the set
    // rows and skip the rest, whose candidates hn / cn and pp = pred_proj(h') [+ bias] stay where the last computation left them.  After a blank
    // tdt_decide itself forms the next step's joint activation z = relu(enc_proj[t'] + pp) for the new frame t' (the one line of SK_ACT's
    // epilogue that depends on t).
    int *need = nullptr;            // [B]
    const float *pp = nullptr;      // [B][J] fp32, natural columns
    const float *ep = nullptr;      // enc_proj [B][T][J]
    float *z = nullptr;             // joint activation [B][J]: fp32 sigma layout, or bf16 natural when h_bf16
    int J = 0;
    // Frame window (small lock-step batches, plain greedy step, prediction-net caching on; 1 = off): the joint is evaluated for the frames t .. t + F - 1 of every
    // utterance in ONE heads product (rows b * F + f of z and logits; the 16-row MFMA tile is mostly empty at B <= 8), and tdt_decide walks through the blank
    // decisions whose successor frame lies inside the window -- the same evaluations in the same order, a run of blanks in one launch instead of one each.
    int F = 1;
    // ragged batches: utterance b has Tb[b] frames, its enc_proj rows start at row0[b] (null: T frames from row b * T); the safety cap on joint
    // evaluations is then per utterance, Tb[b] * (max_symbols + 1) + 16 -- what a single-clip run of that utterance would use
    const int *Tb = nullptr, *row0 = nullptr;
    // Teacher-forced scoring (pk_tdt_score: ONE utterance; pk_stream_score: every stream of a lock-step chunk): the decision of step k is
    // GIVEN -- label force_label[k], duration durations[force_dur[k]] -- instead of taken from the argmax, and the step's joint outputs are
    // recorded: score_lab[k][V] label log-probs, score_dur[k][D] duration log-probs.  The utterance is finished after n_force steps (or when
    // the frame pointer leaves it).  Batched form: utterance b's arrays start at element b * force_stride (rows b * force_stride + k of the
    // score arrays) and it walks n_force_b[b] steps (0: nothing to walk, finished at once).
    const int *force_label = nullptr, *force_dur = nullptr;
    int n_force = 0;
    float *score_lab = nullptr, *score_dur = nullptr;
    const int *n_force_b = nullptr;
    int force_stride = 0;
};
constexpr int kDecWindowMax = 8;    // TdtState::F <= this
constexpr int kMaxListRows = 2048;  // largest lock-step batch the compacted launches handle (larger batches run every row)
void launch_tdt_init(const TdtState &st, hipStream_t s);
// Tb_out[i] / row0_out[i], i < n: frames and first enc_proj row (row_base + its offset) of the utterances of one run; T == nullptr: a uniform
// run of T_uniform frames each
void launch_rag_decode_tables(const int *T, const int *T_off, int T_uniform, int n, int row_base, int *Tb_out, int *row0_out, hipStream_t s);
void launch_lstm_cell(const float *gi, int gi_ld, const int *gi_row, const float *gh, const float *c, int B, int Hp, float *hn,
                      float *cn, hipStream_t s);
void launch_joint_act(const float *ep, const int *t, int T, int J, const float *pp, const float *bp, int B, float *z, hipStream_t s);
void launch_tdt_decide(const TdtState &st, hipStream_t s);

// skinny products of the decode loop (kernels/decode_gemv.hip).  X and W are in the "sigma" K layout
// (4x4 index transpose inside every block of 16 k); K % 16 == 0.
enum SkinnyEpi { SK_BIAS = 0, SK_ACT = 1, SK_CELL = 2 };
struct SkinnyArgs {
    const float *X, *W;            // [B][K], [N][K]
    int B, N, K;
    const float *bias;             // SK_BIAS: [N] or null ; SK_ACT: pred_proj bias (switch A5) or null
    float *out; int ldo;           // SK_BIAS: [B][ldo] natural ; SK_ACT: z [B][N] sigma ; SK_CELL: h' [B][Hp] sigma
    // SK_ACT
    const float *ep; const int *t; int T;
    int F = 1;                     // SK_ACT frame window (TdtState::F): z rows b * F + f, f < F <= kDecWindowMax
    // SK_CELL (N = 4*Hp)
    const float *gi; int gi_ld; const int *gi_row; const float *c; float *cn; int Hp;
    // SK_CELL of an upper LSTM layer with the input projection fused in: gi = X2 W2^T + bias2 (X2 [B][K] sigma, W2 [4Hp][K] sigma); W2 null: gi is read
    const float *X2 = nullptr, *W2 = nullptr, *bias2 = nullptr;
    int nt_weights = 0;            // stream W with non-temporal loads (decode_dev.hpp: NTW)
    // prediction-net caching (TdtState::need): the launch covers only the rows b with need[b] != 0, compacted in ascending order
    // (B <= kMaxListRows); SK_ACT also stores pred_proj(h') [+ bias] to pp_out [B][N]
    const int *need = nullptr;
    float *pp_out = nullptr;
    const int *Tb = nullptr, *row0 = nullptr;   // SK_ACT on a ragged batch (TdtState::Tb / row0)
};
void launch_skinny_gemm(const SkinnyArgs &a, int epi, hipStream_t s);
// the same products in the tolerance-class mode: X / W / X2 / W2 point to bf16 data in NATURAL k order, the SK_ACT / SK_CELL outputs (z, h') are
// bf16; K % 32 == 0 (kernels/decode_gemv_bf16.hip)
void launch_skinny_gemm_bf16(const SkinnyArgs &a, int epi, hipStream_t s);

// the whole greedy loop of a batch in one launch (kernels/decode_persist.hip): the step-invariant arguments of every phase
struct TdtPersist {
    TdtState st;
    int L;
    SkinnyArgs cell[4], ih[4];      // per LSTM layer: W_hh product + cell ; (l > 0) W_ih product of the layer below's h'
    SkinnyArgs act, heads;          // joint activation ; label (+ duration) heads
    unsigned *bar;                  // grid-barrier words, kTdtBarrierWords of them, zeroed before the launch (decode_persist.hip)
    int *abort;                     // set by a workgroup whose barrier wait timed out
    long long timeout_ticks;        // wall_clock64 ticks (100 MHz)
};
constexpr int kTdtBarrierWords = 64 + 8 * 96;
size_t tdt_persistent_lds_bytes(const TdtState &st);
void launch_tdt_persistent(const TdtPersist &p, hipStream_t s);

// ---- LayerNorm, canonical reductions, math diagnostics ------------------------------------------
void launch_layernorm(const float *x, int64_t rows, int d, const float *g, const float *b, float eps, float *y, hipStream_t s, int y_bf16 = 0);   // y_bf16 = 1: y is a bf16 buffer (RNE); 2: fp32, columns in the sigma layout (GemmArgs::a_sigma)
// y1 = LN(x; g1, b1), y2 = LN(y1; g2, b2) in one pass (y1 may alias x)
void launch_layernorm2(const float *x, int64_t rows, int d, const float *g1, const float *b1, const float *g2, const float *b2, float eps,
                       float *y1, float *y2, hipStream_t s, int y2_bf16 = 0);
// Statistics only (round 6): stats[row] = {mean, rstd} of x's rows -- layernorm_kernel's reductions (same sum64 butterflies, same divisions), no output tensor:
// the consumer normalises while it stages the rows (GemmArgs::ln_stats).
void launch_layernorm_stats(const float *x, int64_t rows, int d, float eps, float *stats, hipStream_t s);
// y1 = LN(x; g1, b1) written out (may alias x) + the statistics of y1's rows (what launch_layernorm2's second pass would start from)
void launch_layernorm_then_stats(const float *x, int64_t rows, int d, const float *g1, const float *b1, float eps, float *y1, float *stats, hipStream_t s);
void launch_sum64_rows(const float *x, int rows, int n, float *out, hipStream_t s);
void launch_math(int fn, const float *in, float *out, int64_t n, hipStream_t s);   // 0 exp 1 log 2 tanh 3 sigmoid 4 silu 5 sqrt 6 recip 7 relu
void launch_math_exhaustive(int fn, unsigned long long *out3, hipStream_t s);   // out3 must hold {0, 0, 1 << 32} on entry
void launch_scale(float *x, int64_t n, float a, hipStream_t s);

}  // namespace pk
