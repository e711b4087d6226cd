/*
 * include/parakeet_amd.h -- C ABI of the MI355X-native Parakeet hot path
 * (libparakeet_amd.so, built from parakeet.cpp_amd/csrc by hipcc for gfx950).
 *
 * The reference (Frikallo/parakeet.cpp) has NO plugin / FFI interface: its
 * boundary is the header-only C++ class API in include/parakeet/transcribe.hpp,
 * which calls axiom::Tensor methods directly; a flat C API is an unchecked
 * roadmap item (README.md:518).  This header is therefore the C ABI that the
 * reference's classes would bind if they delegated their arithmetic: each entry
 * point cites the reference interface it replaces.  The source-compatible C++
 * facade (parakeet::Transcriber, TDTTranscriber, TranscribeResult, ...) that sits
 * on top of it lives in parakeet.cpp_amd/include/parakeet/ ; INTEGRATION.md shows
 * the binding a reference maintainer would add.
 *
 * Conventions: plain C types only; the caller owns every input buffer; outputs
 * are caller-allocated unless the function returns an opaque handle, which the
 * library owns until the matching *_free.  Every function returns PK_OK (0) or
 * a negative pk_status; pk_last_error() holds the message (thread-local).
 * Host pointers unless a parameter is named dev_*.  One pk_model may be used
 * from one thread at a time (the reference's objects are not thread-safe
 * either: vocab.hpp:33-35).  There is no CPU fallback: every compute entry point
 * fails with PK_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef PARAKEET_AMD_H
#define PARAKEET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int pk_status;
enum {
    PK_OK = 0,
    PK_ERR_INVALID = -1,     /* bad argument / shape */
    PK_ERR_IO = -2,          /* unreadable weights / vocab / audio (reference: std::runtime_error, vocab.cpp:12-14, audio_io.cpp:271-286) */
    PK_ERR_WEIGHTS = -3,     /* missing tensor or wrong shape (the reference loads non-strict, transcribe.hpp:63; we are strict) */
    PK_ERR_NO_DEVICE = -4,   /* no usable gfx950 device / model not on the GPU */
    PK_ERR_HIP = -5,         /* HIP runtime failure */
    PK_ERR_DECODE_CAP = -6,  /* TDT loop hit the safety cap on joint evaluations (the reference has none, tdt.cpp:62-106) */
    PK_ERR_UNSUPPORTED = -7
};

/* include/parakeet/config.hpp:9-95 (EncoderConfig, PredictionConfig, JointConfig, TDTCTCConfig) flattened. */
typedef struct pk_config {
    int32_t mel_bins;              /* EncoderConfig::mel_bins            80 / 128 */
    int32_t subsampling_channels;  /* EncoderConfig::subsampling_channels 256 */
    int32_t hidden_size;           /* EncoderConfig::hidden_size          512 / 1024 */
    int32_t num_layers;            /* 17 / 24 */
    int32_t num_heads;             /* 8 */
    int32_t ffn_intermediate;      /* 2048 / 4096 */
    int32_t conv_kernel_size;      /* 9 */
    int32_t vocab_size;            /* joint label vocab incl. blank: 1025 / 8193 */
    int32_t pred_hidden;           /* 640 */
    int32_t num_lstm_layers;       /* 1 / 2 */
    int32_t joint_hidden;          /* 640 */
    int32_t num_durations;         /* TDT: 5 ; RNNT head: 0 */
    int32_t durations[8];          /* {0,1,2,3,4} */
    int32_t ctc_vocab_size;        /* 1025, or 0 when the model has no ctc_decoder_ */
    int32_t blank_id;              /* tdt.hpp:71-74 default 1024 ; 600M: vocab_size-1 (main.cpp:252) */
    int32_t max_symbols_per_step;  /* 10 */
    int32_t joint_pred_bias;       /* switch A5: 0 = drop pred_proj_.bias like the reference's Linear(bias=false) (tdt.cpp:10-11) */
    int32_t rnnt_head;             /* 1: joint has a single out_proj_ (rnnt.cpp:37-44) instead of label_/duration_proj_ */
    int32_t stft_window_centered;  /* switch A1: placement of the 400-tap Hann window in the 512-point STFT frame (audio.cpp:117-120;
                                      axiom's stft is not available).  0 (default) = left-aligned, zero-padded on the right: what the
                                      reference author's own check of the C++ features does (scripts/compare_features.py:33-37; pinned by
                                      tests/golden/ref_compare_features_seed7.npz).  1 = centred like torch.stft / NeMo. */
    int32_t gemm_bf16;             /* 0: every product is an fp32 fma chain (bit-identical to the CPU oracle).  1: the encoder-side Linear /
                                      1x1-conv products (and the CTC / enc_proj heads) take bf16 operands with fp32 accumulation on
                                      v_mfma_f32_32x32x16_bf16 -- the precision BASELINE configs[2] names for tdt-600m; the decode loop,
                                      attention scores, norms and depthwise convs stay fp32.  Needs every such K % 64 == 0.  This mode is
                                      compared with the oracle within a tolerance, never bit for bit; since round 2 it also stores the
                                      activations that exist only as GEMM operands as bf16 (the same rounded values) and evaluates SiLU /
                                      sigmoid / the attention softmax on the hardware exp2 / rcp (1 ulp fp32).  Since round 3 the offline
                                      attention (head sizes 64 / 128) and the decode loop's GEMVs also take bf16 operands
                                      (kernels/attention_bf16.hip, decode_gemv_bf16.hip); the specification of the mode is the oracle's
                                      gemm_bf16 mode.  The streaming path (pk_stream_*) and pk_transformer_* keep fp32
                                      attention and exact activations in this mode: only their Linear products and decode GEMVs change.
                                      Round 4: on a streaming model this IS the tolerance-class streaming mode -- every Linear / 1x1-conv
                                      product of a chunk on bf16 operands (kernels/gemm_smallm_bf16.hip: K split over the waves of a workgroup,
                                      the LayerNorm of a product's input folded in), specification = the oracle's Stream with gemm_bf16 = 1,
                                      compared within the bounds of tests/test_gpu_stream.py; 16 lock-step streams of
                                      nemotron-600m cost 2.4-2.5 ms per 160 ms chunk instead of 3.1-3.2. */
    char joint_prefix[32];         /* "tdt_joint_." (tdt_ctc.cpp:5-9) or "joint_." (tdt.cpp:28-32) */
    /* encoder-only uses (Sortformer's NEST encoder, src/sortformer.cpp:41-47): vocab_size = 0 loads no prediction net / joint */
    int32_t xscaling;              /* StreamingEncoderConfig::xscaling (streaming_encoder.cpp:402-406, :444-447): x *= sqrt(hidden) after subsampling */
    int32_t mel_normalize_off;     /* AudioConfig::normalize = false (audio.cpp:140): features are the raw log-mel (Sortformer, main.cpp:516) */
    char encoder_prefix[32];       /* module name of the encoder in the state dict; "" = "encoder_." ; Sortformer: "nest_encoder_." */
} pk_config;

/* make_110m_config / make_tdt_600m_config / make_rnnt_600m_config (config.hpp:77-135). name: "tdt-ctc-110m" | "tdt-600m" | "rnnt-600m" |
 * "nemotron-600m" (nemotron.hpp:31-52) | "eou-120m" (eou.hpp:34-56) -- streaming models, see pk_stream_*. */
pk_status pk_config_preset(const char *name, pk_config *out);

typedef struct pk_model pk_model;

const char *pk_version(void);
/* Copies the calling thread's last error message; returns its length. */
size_t pk_last_error(char *buf, size_t cap);
int pk_device_count(void);

/* ---- model lifetime: Transcriber::Transcriber (transcribe.hpp:59-65), to_gpu() (:68-71) ------------------ */
/* safetensors with the reference's tensor names (scripts/convert_nemo.py:98-310); vocab_path may be NULL. */
pk_status pk_model_load(const char *safetensors_path, const char *vocab_path, const pk_config *cfg, pk_model **out);
/* The same from a safetensors image in memory (copied): what a rank receives when rank 0 reads the file once and broadcasts it
 * (RCCL) instead of every rank going to storage. */
pk_status pk_model_load_buffer(const void *safetensors_image, size_t n_bytes, const char *vocab_path, const pk_config *cfg, pk_model **out);
/* Upload (packed) weights to HBM on `device` and build the execution plan.  Idempotent per device.
 * HBM use: the weights once, plus -- fp32 mode, on the first call that runs fewer than 1537 encoder rows at a time (one or a few clips,
 * every streaming session) -- a second copy of the encoder's Linear weights tiled for the small-batch kernels (0.4 GB for tdt-ctc-110m,
 * 2.4 GB for the 600M models), shared by all later calls and streams of the model. */
pk_status pk_model_to_gpu(pk_model *m, int device);
void pk_model_free(pk_model *m);
pk_status pk_model_config(const pk_model *m, pk_config *out);
/* How the TDT / RNNT greedy loop (src/tdt.cpp:62-106) is issued.  Every mode runs the SAME device functions and gives identical results
 * (tests/test_gpu_decode.py); they differ in launch structure only.  PHASES (default): one launch per phase of a symbol step (LSTM cells,
 * joint activation, heads, decision), the host polls a done-counter every 16 steps.  PERSISTENT: the whole loop in one launch, phases
 * separated by a grid barrier (needs <= 2 LSTM layers, no phrase boosting, no carried streaming state; otherwise PHASES is used).
 * GRAPH: the 16-step chunk of PHASES captured once as a hipGraph and replayed. */
enum { PK_DECODE_LOOP_PHASES = 0, PK_DECODE_LOOP_PERSISTENT = 1, PK_DECODE_LOOP_GRAPH = 2 };
pk_status pk_model_set_decode_loop(pk_model *m, int mode);

/* ---- stage entry points (host buffers; used by the parity tests and the C++ facade) ----------------------- */
/* preprocess_audio (src/audio.cpp:100-158): n_clips clips of n_samples each -> feats[n_clips][n_frames][mel_bins],
 * n_frames = 1 + n_samples/160.  logmel (optional, may be NULL): [n_clips][mel_bins][n_frames] before normalisation. */
pk_status pk_mel(pk_model *m, const float *pcm, int n_clips, int64_t n_samples, float *feats, float *logmel);
int pk_mel_num_frames(int64_t n_samples);
int pk_encoder_num_frames(int n_mel_frames); /* floor((n-1)/2)+1 three times (encoder.cpp:208-217) */

/* FastConformerEncoder::forward (src/encoder.cpp:253-271): feats[B][Tm][mel] -> enc[B][T][hidden].
 * stop_layer / stop_stage cut the pipeline for stage-wise parity checks: run `stop_layer` full blocks, then the
 * next block up to stop_stage (0 none, 1 ffn1, 2 +attn, 3 +conv, 4 +ffn2); pass (num_layers, 0) -- or (-1, 0) -- for all. */
pk_status pk_encode(pk_model *m, const float *feats, int B, int Tm, int stop_layer, int stop_stage, float *enc);
/* ConformerBlock::forward x n_layers starting at first_layer (src/encoder.cpp:196-204, the loop at :267-269), on an encoder
 * stream x[B][T][hidden] given by the caller.  This is the boundary the reference author's own PyTorch check
 * (scripts/compare_encoder.py) cuts at: tests/golden/ holds its outputs. */
pk_status pk_conformer_blocks(pk_model *m, const float *x_in, int B, int T, int first_layer, int n_layers, float *x_out);
/* ConvSubsampling::forward only (src/encoder.cpp:219-241). */
pk_status pk_subsample(pk_model *m, const float *feats, int B, int Tm, float *out);

/* CTCDecoder::forward + ctc_greedy_decode(_with_timestamps) (src/ctc.cpp:12-25, :40-127).
 * ids/start/end/conf: [B][T] (NULL to skip the optional ones); lens[B]; logp (optional) [B][T][ctc_vocab]. */
pk_status pk_ctc_decode(pk_model *m, const float *enc, int B, int T, int32_t *ids, int32_t *lens, int32_t *start,
                        int32_t *end, float *conf, float *logp);
/* tdt_greedy_decode(_with_timestamps) (src/tdt.cpp:36-201; RNNT head: src/rnnt.cpp:56-177).
 * ids/start/end/conf: [B][max_tokens]; lens[B] (-1 if the safety cap hit); steps[B] joint evaluations (optional). */
pk_status pk_tdt_decode(pk_model *m, const float *enc, int B, int T, int max_tokens, int32_t *ids, int32_t *lens,
                        int32_t *start, int32_t *end, float *conf, int32_t *steps);

/* ---- ragged (mixed-length) forms of the stage entry points -------------------------------------------------------------------
 * The reference's roadmap item "Batch inference: pad + length-mask multiple audio files, batch through encoder and decoder" (README.md:513;
 * mask seam src/encoder.cpp:163-165) -- done by PACKING instead of padding: the B clips of a batch lie back to back along the time axis of
 * every tensor, GEMMs / LayerNorm / activations are row-wise and see one tall matrix, and the kernels that look across rows (STFT framing,
 * the stride-2 convolutions, the depthwise conv, attention with its relative-position table, the greedy decoders) take per-clip extents.
 * No padded frame is ever computed and no mask is needed; every clip's result is BIT-IDENTICAL to the same clip run alone (same fma
 * chains, same softmax extent, same zero padding at its own edges) -- tests/test_gpu_ragged.py.
 * pk_mel_ragged: clip i = pcm[offsets[i] .. offsets[i+1]) -> feats packed [sum_i Tm_i][mel_bins], Tm_i = pk_mel_num_frames(len_i); logmel
 *   (optional) per clip one [mel_bins][Tm_i] block, blocks back to back.
 * pk_encode_ragged: feats packed as above, n_mel_frames[B] -> enc packed [sum_i T_i][hidden], T_i = pk_encoder_num_frames(Tm_i).
 * pk_conformer_blocks_ragged: x packed [sum_i n_frames[i]][hidden] in and out.
 * pk_ctc_decode_ragged / pk_tdt_decode_ragged: enc packed, n_frames[B]; token arrays [B][max_i n_frames[i]] resp. [B][max_tokens] as in the
 *   uniform calls; logp (optional) packed [sum_i T_i][ctc_vocab]. */
pk_status pk_mel_ragged(pk_model *m, const float *pcm, const int64_t *offsets, int n_clips, float *feats, float *logmel);
pk_status pk_encode_ragged(pk_model *m, const float *feats, const int32_t *n_mel_frames, int B, int stop_layer, int stop_stage, float *enc);
pk_status pk_conformer_blocks_ragged(pk_model *m, const float *x_in, const int32_t *n_frames, int B, int first_layer, int n_layers, float *x_out);
pk_status pk_ctc_decode_ragged(pk_model *m, const float *enc, const int32_t *n_frames, int B, int32_t *ids, int32_t *lens, int32_t *start,
                               int32_t *end, float *conf, float *logp);
pk_status pk_tdt_decode_ragged(pk_model *m, const float *enc, const int32_t *n_frames, int B, int max_tokens, int32_t *ids, int32_t *lens,
                               int32_t *start, int32_t *end, float *conf, int32_t *steps);

/* Teacher-forced joint scores -- "TDT logits within stated fp tolerance" made checkable for a greedy decoder: the loop of
 * tdt_greedy_decode (src/tdt.cpp:62-106) on ONE utterance enc[T][hidden] with the decision of every step GIVEN (labels[k], and
 * dur_idx[k] = an index into pk_config.durations) instead of taken from the argmax, so that the state every step is scored in -- frame
 * pointer, last token, LSTM state -- is the one another implementation (the CPU oracle, the reference) was in at the same step.  A blank
 * reverts the LSTM state and advances by max(duration, 1) (:88-93); a token commits it and advances by its duration (0: same frame,
 * :95-105).  Outputs per step k < *n_done: label_logp[k][vocab_size], dur_logp[k][num_durations] = the joint's log-softmax outputs
 * (TDTJoint::forward, :15-24); either may be NULL.  *n_done = steps evaluated (< n_steps when the frame pointer left the utterance). */
pk_status pk_tdt_score(pk_model *m, const float *enc, int T, const int32_t *labels, const int32_t *dur_idx, int n_steps, float *label_logp,
                       float *dur_logp, int *n_done);

/* Early warning of the tolerance-class (bf16) mode, SURVEY.md 8(c): per utterance of the LAST pk_tdt_decode on this model, the smallest
 * (top-1 minus top-2) log-prob over all of its greedy decisions -- the label argmax (tdt.cpp:78-82) and, for TDT heads, the duration argmax
 * (:84-86: a flip there moves the frame pointer and every later token with it): how close the decode came to a different path.  A margin below the mode's numerical error marks a token that may differ from the reference.  Not produced for
 * boosted or streaming decodes. */
pk_status pk_decode_margins(pk_model *m, float *min_margin, int B);

/* ---- resident batch pipeline (device buffers; what bench.py times) ---------------------------------------- */
typedef struct pk_batch pk_batch;
enum { PK_DECODER_CTC = 0, PK_DECODER_TDT = 1 }; /* enum class Decoder (transcribe.hpp:34) */
/* A batch slot of up to max_clips clips of exactly n_samples samples, all buffers resident in HBM. */
pk_status pk_batch_create(pk_model *m, int max_clips, int64_t n_samples, pk_batch **out);
void pk_batch_free(pk_batch *b);
/* The same pipeline with RAGGED capacity: every run takes up to max_clips clips of ANY lengths (<= max_clip_samples each, <=
 * max_total_samples in all), packed -- see the ragged stage entry points above.  pk_batch_upload_ragged: clip i = pcm[offsets[i] ..
 * offsets[i+1]); a batch whose clips all have one length runs the uniform kernels.  pk_batch_run / _sync / _results* / decode groups work
 * as for uniform pipelines; the token arrays are [n_clips][pk_batch_max_tokens], pitched for the longest clip the pipeline can hold. */
pk_status pk_batch_create_ragged(pk_model *m, int max_clips, int64_t max_total_samples, int64_t max_clip_samples, pk_batch **out);
pk_status pk_batch_upload_ragged(pk_batch *b, const float *pcm, const int64_t *offsets, int n_clips);
pk_status pk_batch_upload_ragged_async(pk_batch *b, const float *pcm, const int64_t *offsets, int n_clips);
/* Host -> HBM copy of the PCM (outside bench's timed region). */
pk_status pk_batch_upload(pk_batch *b, const float *pcm, int n_clips);
/* mel -> encoder -> decode, enqueued on the batch's stream; returns without synchronising. */
/* Streams of distinct batches: copy the NEXT batch into the second PCM buffer on a copy stream while the current encoder runs
 * (no flush; the host buffer may be reused after the call returns for pageable memory, after the next call into the batch for
 * pinned memory).  The next pk_batch_run consumes it. */
pk_status pk_batch_upload_async(pk_batch *b, const float *pcm, int n_clips);
pk_status pk_batch_run(pk_batch *b, int decoder);
pk_status pk_batch_sync(pk_batch *b);
/* Token ids etc. of the last run (synchronises).  Arrays [n_clips][max_tokens]; max_tokens = pk_batch_max_tokens. */
int pk_batch_max_tokens(const pk_batch *b);
pk_status pk_batch_results(pk_batch *b, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf);
/* Results of the newest batch whose decode has FINISHED -- run k's decode is driven inside pk_batch_run(k+1) -- without flushing the
 * decode still pending: the consumer side of the pipeline (run(k+1); results_done -> batch k).  *n_clips = its clip count. */
pk_status pk_batch_results_done(pk_batch *b, int *n_clips, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf);
/* Decode groups (throughput mode of the pipeline; default 1 = decode(k) under encoder(k+1)).  With group = G the TDT / RNNT greedy loops of
 * G consecutive pk_batch_run calls are driven as ONE lock-step batch of G * n_clips utterances under the encoder of the run after them:
 * the loop of tdt_greedy_decode (src/tdt.cpp:62-106) is launch-bound -- four launches per symbol step whatever the batch -- so G runs
 * share them.  Token ids, frames and confidences of every run are unchanged (the utterances are independent); what changes is WHEN they
 * are available: after the G-th run of the group (+1), or at pk_batch_sync / pk_batch_results.  1 <= G <= 16; flushes the pipeline. */
pk_status pk_batch_set_decode_group(pk_batch *b, int group);
/* on (default): the decode loop of run k / of a finished group runs on a second, high-priority stream UNDER the encoder of the following run.
 * off: it runs on the encoder's stream, after the encoder -- no concurrency on the device.  The decode GEMVs stream the prediction-net and
 * joint weights through L2 once per symbol step (13 MB for tdt-ctc-110m, 42 MB for tdt-600m); for the large heads that traffic costs the
 * concurrently running encoder GEMMs more than the loop's own duration, so serial issue can be the faster schedule.
 * Results are identical either way.  Flushes the pipeline. */
pk_status pk_batch_set_decode_overlap(pk_batch *b, int on);
/* Results of the (back+1)-th newest run whose decode has finished (back = 0: the newest, = pk_batch_results_done), 0 <= back <
 * pk_batch_results_available().  A finished run stays readable until its buffers are recycled: without groups until the run after next is
 * issued, with groups of G until the first run of the group after next -- so the G runs of the newest decoded group are always there, and a
 * pk_batch_sync that decodes a full group and the partial group behind it keeps the runs of both.  pk_batch_set_decode_group drops the
 * runs held in group buffers: read them first. */
pk_status pk_batch_results_back(pk_batch *b, int back, int *n_clips, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf);
int pk_batch_results_available(const pk_batch *b);
/* pk_decode_margins for the (back+1)-th newest finished run of the pipeline: min_margin[n_clips of that run]. */
pk_status pk_batch_margins(pk_batch *b, int back, float *min_margin);
/* Stage timers of the last pk_batch_run_timed (ms): mel, encoder, decode, total (hipEvents on the batch stream). */
pk_status pk_batch_run_timed(pk_batch *b, int decoder, float ms[4]);
/* Raw device pointers for zero-copy producers (e.g. torch tensors): PCM [max_clips][n_samples] f32. */
void *pk_batch_dev_pcm(pk_batch *b);
void *pk_batch_stream(pk_batch *b);

/* Per-kernel timing of ONE pk_batch_run with hipEvents around every launch (slow; for the roofline object only).
 * Fills up to cap records; returns the number of distinct kernels. */
typedef struct pk_kernel_stat {
    char name[64];
    int32_t launches;
    float total_ms;
    double flops;  /* algorithmic flops of those launches (0 for memory-bound kernels) */
    double bytes;  /* algorithmic HBM bytes of those launches */
} pk_kernel_stat;
int pk_batch_profile(pk_batch *b, int decoder, pk_kernel_stat *out, int cap);

/* ---- one-call API: Transcriber::transcribe (transcribe.hpp:74-179) ---------------------------------------- */
typedef struct pk_options {     /* TranscribeOptions (transcribe.hpp:38-43) */
    int32_t decoder;            /* PK_DECODER_* */
    int32_t timestamps;
    /* boost_phrases / boost_score (transcribe.hpp:41-42): when n_boost_phrases > 0 the phrases are tokenised with the model's
     * vocabulary (ContextTrie::build, src/phrase_boost.cpp:29-37) and this call decodes boosted; 0 phrases = the model-level
     * setting of pk_set_boost_* (off by default).  boost_score is used as given (the reference's default is 5.0). */
    const char *const *boost_phrases;
    int32_t n_boost_phrases;
    float boost_score;
} pk_options;
typedef struct pk_word {        /* WordTimestamp (timestamp.hpp:18-23) */
    const char *word;
    float start, end, confidence;
} pk_word;
typedef struct pk_result {      /* TranscribeResult (transcribe.hpp:23-30) + TimestampedToken (timestamp.hpp:11-16) */
    const char *text;
    int32_t n_tokens;
    const int32_t *token_ids;
    const int32_t *start_frame; /* NULL unless options.timestamps */
    const int32_t *end_frame;
    const float *confidence;
    int32_t n_words;
    const pk_word *words;
} pk_result;
/* Clips are pcm[offsets[i] .. offsets[i+1]), of ANY lengths: they are sorted by length and packed into ragged batches (<= 256 clips, <= 8192
 * encoder rows = 655 s of audio per batch: the row count at which every GEMM of the encoder fills whole rounds of the 256 CUs) that go through
 * the two-stream pipeline; every clip's result is bit-identical to transcribing it alone.
 * results: array of n_clips pk_result, owned by the library until pk_results_free. */
pk_status pk_transcribe_pcm(pk_model *m, const float *pcm, const int64_t *offsets, int n_clips, const pk_options *opt,
                            pk_result **results);
/* The packing policy of pk_transcribe_pcm / pk_group_transcribe_pcm on its own (host logic, no GPU needed): clips of n_samples[i] samples
 * are sorted by length (longest first, stable) and cut into batches of <= 256 clips and <= 8192 encoder rows (pk_encoder_num_frames of
 * pk_mel_num_frames of the length; 65 clips of 10 s); batch_of_clip[i] = the
 * batch clip i lands in (batch 0 holds the longest clips), pos_in_batch[i] (optional) = its row in that batch. */
pk_status pk_plan_batches(const int64_t *n_samples, int n_clips, int32_t *batch_of_clip, int32_t *pos_in_batch, int *n_batches);
/* Per-clip extents of a ragged batch (host logic): mel frames pk_mel_num_frames(n), encoder frames pk_encoder_num_frames(...) of every
 * clip; totals (optional, 7 values): samples, mel frames, rows after the second stride-2 stage, encoder rows, attention row blocks,
 * depthwise-conv strips, subsampling strips of the packed batch -- what the ragged kernels' grids are sized by. */
pk_status pk_ragged_extents(const int64_t *n_samples, int n_clips, int32_t *n_mel_frames, int32_t *n_enc_frames, int64_t *totals);
void pk_results_free(pk_result *results, int n_clips);
/* ---- one node, several GPUs: utterance shards (the reference has no multi-device path, README.md:513) ----------
 * A pk_group is one model REPLICA per device of this process: the safetensors file is mapped once and every replica is built from that
 * one host image by its own host thread (each device uploads over its own PCIe link).  pk_group_transcribe_pcm deals the clips, longest
 * first, each to the device with the least audio so far (equal lengths: rank r takes clips r, r+G, ...); every device packs its clips into
 * ragged batches; one host thread drives each device through the SAME two-stream pipeline pk_transcribe_pcm uses (PCM of batch k+1 staged and decode(k) driven
 * under encoder(k+1); from four batches per rank on, decode groups of four).  Utterances share nothing, so there is NO collective: not on
 * the data path and -- one process, one address space -- not for the results either; the ranks never wait for each other.  The library has
 * no link-time dependency on RCCL.  (The multi-PROCESS deployment -- one process per GPU under torch.distributed, bench.py --gpus N,
 * tools/transcribe_sharded.py -- ends with one RCCL all-gather of the token matrix; pk_group_verify_exchange below runs that exchange
 * in-process as a check.)  devices = NULL / n_devices <= 0: every visible device.
 * Same result contract as pk_transcribe_pcm (which is what a group of one device computes, clip for clip). */
typedef struct pk_group pk_group;
pk_status pk_group_create(const char *safetensors_path, const char *vocab_path_or_null, const pk_config *cfg, const int *devices,
                          int n_devices, pk_group **out);
void pk_group_free(pk_group *g);
int pk_group_size(const pk_group *g);
pk_status pk_group_transcribe_pcm(pk_group *g, const float *pcm, const int64_t *offsets, int n_clips, const pk_options *opt,
                                  pk_result **results);
/* figures of the last pk_group_transcribe_pcm: max-over-ranks wall time of the compute phase, total audio seconds, clips handled by
 * each rank (clips_per_rank[pk_group_size]); any pointer may be NULL */
pk_status pk_group_last_stats(const pk_group *g, double *wall_ms_max, double *audio_seconds, int32_t *clips_per_rank);
/* Debug check of the result exchange a multi-process deployment performs, run in-process over RCCL (loaded with dlopen on first use;
 * PK_ERR_UNSUPPORTED with the loader's message when librccl is not installed): the token ids of `results` (those of the LAST
 * pk_group_transcribe_pcm) go rank by rank through device memory, one ncclAllReduce(max) of (token maximum, wall time) and one fixed-stride
 * ncclAllGather of the [clips_per_rank][2 + max_tokens] int32 matrix; every rank's gathered copy must reproduce `results`.
 * *rccl_ranks (optional) = ncclCommCount of the communicator. */
pk_status pk_group_verify_exchange(pk_group *g, const pk_result *results, int n_clips, int *rccl_ranks);

/* read_audio (audio_io.cpp:453-483) restricted to RIFF/WAVE PCM16 / float32; mono-downmix; must be 16 kHz.
 * Returns a malloc'd buffer the caller frees with pk_free. */
pk_status pk_read_wav(const char *path, float **pcm, int64_t *n_samples, int *sample_rate);
/* read_audio(path, target_sample_rate) (audio_io.cpp:453-483) for RIFF/WAVE files: decode, mono downmix (sum * 1/channels,
 * :198-214) and resampling to target_rate with the reference's Kaiser-windowed sinc interpolator (sinc_resample, :123-195; fp64,
 * beta 7.857, 16-tap half width).  FLAC / MP3 / OGG are not decoded.  pk_resample: the resampler alone (resample(), :250-262). */
pk_status pk_read_audio(const char *path, int target_rate, float **pcm, int64_t *n_samples, int *original_rate);
pk_status pk_resample(const float *pcm, int64_t n, int src_rate, int dst_rate, float **out, int64_t *n_out);
/* read_audio(const uint8_t *data, size_t len, target) (audio_io.hpp:27-28): an encoded RIFF/WAVE image in memory -> mono PCM at
 * target_rate (malloc'd, pk_free). */
pk_status pk_read_audio_memory(const void *data, size_t len, int target_rate, float **pcm, int64_t *n_samples, int *original_rate, int *n_channels);
/* get_audio_duration (audio_io.hpp:39, audio_io.cpp:527-586): header walk, no decode.  duration = n_frames / sample_rate. */
pk_status pk_audio_info(const char *path, int *sample_rate, int *n_channels, int64_t *n_frames);
void pk_free(void *p);

/* ---- streaming: NemotronTranscriber / StreamingTranscriber::transcribe_chunk (src/nemotron.cpp:24-52, src/eou.cpp:113-146) -------- */
/* A pk_stream is n_streams independent streaming sessions advanced in LOCK-STEP on one GPU (BASELINE configs[4]: 16 concurrent
 * streams per GPU): every push hands EACH stream the same number of samples.  The model is a TDT-joint model loaded with
 * pk_model_load / pk_model_to_gpu (the streaming encoder uses the offline encoder's tensor names, streaming_encoder.cpp:276-428).
 * State kept per stream: pre-emphasis carry + overlap samples (StreamingAudioPreprocessor, audio.cpp:171-259), leftover mel
 * frames, per-layer K/V and conv caches (EncoderCache, streaming_encoder.hpp:25-41), LSTM state + last token + frame offset
 * (StreamingDecodeState, eou.hpp:80-87).  att_context_left / right: StreamingEncoderConfig (streaming_encoder.hpp:17-23;
 * Nemotron 70 / latency_frames, EOU 70 / 1).  pk_config.xscaling is honoured (the Sortformer NEST preset sets it:
 * streaming_encoder.cpp:444-447); the SiLU subsampling variant of that config is not implemented (no shipped preset enables it). */
typedef struct pk_stream pk_stream;
pk_status pk_stream_create(pk_model *m, int n_streams, int att_context_left, int att_context_right, pk_stream **out);
void pk_stream_free(pk_stream *s);
pk_status pk_stream_reset(pk_stream *s);    /* NemotronTranscriber::reset (nemotron.cpp:54-58) */
/* pcm[n_streams][n_samples] -> the tokens each stream emitted for this chunk: ids/start/end/conf [n_streams][max_tokens]
 * (start / end: encoder frames since the stream began, eou.cpp:77-79), lens[n_streams] (0 while audio is still being buffered).
 * Limit: one push carries at most 79 360 samples (4.96 s = 62 encoder frames: the per-chunk decode workspace); a longer push is rejected
 * with PK_ERR_UNSUPPORTED BEFORE any carried state changes -- split it into several pushes (the reference's chunks are 1-14 frames). */
pk_status pk_stream_push(pk_stream *s, const float *pcm, int n_samples, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start,
                         int32_t *end, float *conf);
/* the three stages of a push, on host buffers, for parity tests (each consumes / updates the same carried state):
 * process_chunk -> out[n_streams][*n_frames][mel_bins] (un-normalised log-mel); forward_chunk -> enc[n_streams][*n_out][hidden];
 * rnnt_streaming_decode_chunk.  *n_frames / *n_out = 0: everything was buffered. */
pk_status pk_stream_mel(pk_stream *s, const float *pcm, int n_samples, float *out, int cap_frames, int *n_frames);
pk_status pk_stream_encode(pk_stream *s, const float *mel, int n_frames, float *enc, int cap_frames, int *n_out);
pk_status pk_stream_decode(pk_stream *s, const float *enc, int n_frames, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start,
                           int32_t *end, float *conf);
/* rnnt_streaming_decode_chunk (src/eou.cpp:17-98) of every stream along a GIVEN decision path -- the streaming counterpart of pk_tdt_score
 * (parity tests of the tolerance-class mode: "TDT logits within stated fp tolerance").  enc[n_streams][n_frames][hidden]; stream s walks
 * n_steps[s] decisions labels[s][k] / dur_idx[s][k] (both [n_streams][cap]; a duration as an index into pk_config.durations) instead of its
 * argmax, and the joint's outputs of every step (TDTJoint::forward, src/tdt.cpp:15-24) are recorded: label_logp[n_streams][cap][vocab]
 * (may be NULL), dur_logp[n_streams][cap][num_durations] (may be NULL); rows beyond a stream's steps are zero.  The LSTM state, last token
 * and frame offset each stream carries into its next chunk are the ones that path leaves (StreamingDecodeState, eou.hpp:80-87).
 * n_done[n_streams] (may be NULL) = steps walked (= n_steps[s] for a path the reference's loop can take on this chunk). */
pk_status pk_stream_score(pk_stream *s, const float *enc, int n_frames, const int32_t *labels, const int32_t *dur_idx, const int32_t *n_steps,
                          int cap, float *label_logp, float *dur_logp, int32_t *n_done);

/* ---- plain Transformer encoder: TransformerEncoder::forward / TransformerBlock::forward (src/transformer.cpp:15-88) ------------ */
/* include/parakeet/transformer.hpp:12-21 (TransformerConfig), dropout omitted (inference). */
typedef struct pk_transformer_config {
    int32_t hidden_size;       /* 192 */
    int32_t num_layers;        /* 18 */
    int32_t num_heads;         /* 8 (heads narrower than 32 are zero-padded internally: same bits) */
    int32_t ffn_intermediate;  /* 768 */
    int32_t pre_ln;            /* 1 = pre-norm (x + f(LN(x))), 0 = post-norm (LN(x + f(x))) */
    int32_t has_final_norm;
    float layer_norm_eps;      /* 1e-5 */
} pk_transformer_config;
typedef struct pk_transformer pk_transformer;
/* Tensors <prefix>layers_.<i>.{norm1_,norm2_,mha_.{q,k,v,out}_proj,fc1_,fc2_}.{weight,bias} [+ <prefix>final_norm_.*] of a
 * safetensors file (the module tree of transformer.cpp:12,69-73), uploaded to `device`. */
pk_status pk_transformer_load(const char *safetensors_path, const char *prefix, const pk_transformer_config *cfg, int device,
                              pk_transformer **out);
/* x[B][T][hidden] -> y[B][T][hidden] (host buffers).  The optional attention mask of the reference (transformer.cpp:40-42) is
 * not supported: no caller on the ASR path passes one. */
pk_status pk_transformer_forward(pk_transformer *t, const float *x, int B, int T, float *y);
void pk_transformer_free(pk_transformer *t);

/* ---- preprocess_audio on its own (include/parakeet/audio.hpp:7-30, src/audio.cpp:100-158): the mel front end without a model, for callers
 * that hand features to pk_sortformer_forward / pk_encode themselves.  feats: [pk_mel_num_frames(n_samples)][n_mels]; normalize = 0 is
 * AudioConfig::normalize = false (raw log-mel, what Sortformer takes); stft_window_centered: switch A1 of pk_config. */
typedef struct pk_frontend pk_frontend;
pk_status pk_frontend_create(int n_mels, int normalize, int stft_window_centered, int device, pk_frontend **out);
pk_status pk_frontend_features(pk_frontend *f, const float *pcm, int64_t n_samples, float *feats, int *n_frames);
void pk_frontend_free(pk_frontend *f);

/* ---- Sortformer speaker diarization (include/parakeet/sortformer.hpp:28-129, src/sortformer.cpp:41-121) ------------------------
 * NEST FastConformer (offline Conformer path, xscaling, weights under "nest_encoder_.") -> projection_ -> TransformerEncoder
 * ("transformer_.") -> relu -> first_hidden_ -> relu -> output_proj_ -> sigmoid.  One handle = weights on one device. */
typedef struct pk_sortformer_config {   /* SortformerConfig (sortformer.hpp:28-41) */
    pk_config nest;                     /* encoder fields only: vocab_size = ctc_vocab_size = 0, xscaling = 1, mel_normalize_off = 1 */
    pk_transformer_config transformer;
    int32_t max_speakers;               /* 4 */
    float activity_threshold;           /* 0.5 */
    int32_t att_context_left;           /* 70: StreamingEncoderConfig of the NEST encoder (sortformer.hpp:53-54), used by diarize_chunk */
    int32_t att_context_right;          /* 0 */
} pk_sortformer_config;
typedef struct pk_sortformer pk_sortformer;
void pk_sortformer_config_preset(pk_sortformer_config *out);              /* make_sortformer_117m_config (sortformer.hpp:43-76) */
pk_status pk_sortformer_load(const char *safetensors_path, const pk_sortformer_config *cfg, int device, pk_sortformer **out);
void pk_sortformer_free(pk_sortformer *s);
/* Sortformer::forward (:50-69): feats[B][Tm][mel_bins] -> probs[B][T][max_speakers], T = pk_encoder_num_frames(Tm) (also in *T_out). */
pk_status pk_sortformer_forward(pk_sortformer *s, const float *feats, int B, int Tm, float *probs, int *T_out);
/* preprocess_audio with normalize = false (src/main.cpp:513-517, src/diarize.cpp:81-88) + forward, for n_clips clips of n_samples. */
pk_status pk_sortformer_forward_pcm(pk_sortformer *s, const float *pcm, int n_clips, int64_t n_samples, float *probs, int *T_out);
/* Sortformer::diarize_chunk (:123-150), one streaming session per handle: forward_chunk of the NEST encoder on cached K / V / conv
 * state (the streaming path of pk_stream_*), then projection / transformer / head on THIS chunk's frames.  feats[n_frames][mel_bins]
 * (un-normalised log-mel, any chunking) -> probs[c][max_speakers]; *T_out = c (0: all frames were buffered: the subsampling consumes
 * multiples of 8).  AOSCCache::update and probs_to_segments on the chunk are host loops (pk_sortformer_segments). */
pk_status pk_sortformer_diarize_chunk(pk_sortformer *s, const float *feats, int n_frames, float *probs, int cap_frames, int *T_out);
pk_status pk_sortformer_stream_reset(pk_sortformer *s);
/* Sortformer::probs_to_segments (:71-113) on one utterance's probs[T][S]: runs of prob > threshold per speaker, seconds = frame * 0.08,
 * sorted by start.  Writes at most cap segments, returns the number found. */
int pk_sortformer_segments(const float *probs, int T, int S, float threshold, int32_t *speaker, float *start, float *end, int cap);

/* ---- host-side text (src/vocab.cpp:29-117, src/timestamp.cpp:24-111) -------------------------------------- */
int pk_vocab_size(const pk_model *m);
/* Tokenizer::decode -> returns needed length; writes at most cap-1 bytes + NUL. */
int pk_detokenize(const pk_model *m, const int32_t *ids, int n, char *out, int cap);
/* Tokenizer::encode (greedy longest match) -> number of ids (writes at most cap). */
int pk_tokenize(const pk_model *m, const char *text, int32_t *ids, int cap);
/* Phrase boosting (include/parakeet/phrase_boost.hpp:22-116, src/phrase_boost.cpp): a ContextTrie over token sequences biases
 * the CTC / TDT greedy argmax by +boost_score towards tokens that continue a phrase.  The trie lives on the device with the
 * model; once set, pk_ctc_decode, pk_tdt_decode, pk_batch_run and pk_transcribe_pcm decode boosted (confidences stay the
 * unboosted probabilities).  n_phrases = 0 switches boosting off.  Phrases are limited to 63 tokens.  RNNT decode and streaming
 * sessions have no boosted variant in the reference: PK_ERR_UNSUPPORTED while boosting is on.
 * pk_set_boost_tokens: phrase i = ids[offsets[i] .. offsets[i+1])  (ContextTrie::insert, :11-27; empty phrases are ignored).
 * pk_set_boost_phrases: ContextTrie::build (:29-37) -- Tokenizer::encode of each phrase; needs a vocabulary. */
pk_status pk_set_boost_tokens(pk_model *m, const int32_t *ids, const int32_t *offsets, int n_phrases, float boost_score);
pk_status pk_set_boost_phrases(pk_model *m, const char *const *phrases, int n_phrases, float boost_score);
/* number of trie nodes (ContextTrie::size, 1 = root only), 0 when boosting is off */
int pk_boost_trie_size(const pk_model *m);
/* group_timestamps: words '\n'-joined into `words`; returns the word count. sentences!=0 -> TimestampMode::Sentences. */
int pk_group_timestamps(const pk_model *m, const int32_t *ids, const int32_t *start, const int32_t *end, const float *conf,
                        int n, int sentences, char *words, int cap, float *wstart, float *wend, float *wconf, int wcap);

/* ---- diagnostics used by the GPU parity tests (single kernels behind the same ABI) ------------------------ */
/* Device math, elementwise: fn 0 exp, 1 log, 2 tanh, 3 sigmoid, 4 silu, 5 sqrt, 6 reciprocal, 7 relu, 8 / 9 sigmoid / silu as the GEMM
 * epilogues evaluate them (guarded short sequences, pk_devmath.h). */
pk_status pk_diag_math(int fn, const float *in, float *out, int64_t n);
/* Every one of the 2^32 bit patterns of x through a device-side identity (kernels/norm.hip, math_exhaustive_kernel): fn 3 / 4 = wherever the
 * short sigmoid / SiLU instruction sequences of the GEMM epilogues claim validity they equal the specification's value; fn 13 / 14 = the guarded
 * four-at-a-time forms equal the specification on every pattern.  checked = patterns examined, mismatches must be 0, first_bad = lowest
 * mismatching pattern (2^32 when none).  About 0.1 s. */
pk_status pk_diag_math_exhaustive(int fn, uint64_t *checked, uint64_t *mismatches, uint64_t *first_bad);
/* out[M][N] = epi(A[M][K] * W[N][K]^T + bias) with the production fp32-MFMA GEMM.  epi: 0 none, 1 relu, 2 silu,
 * 3 residual: out = resid + alpha*(acc+bias), 4 glu (N even: out[M][N/2] = a * sigmoid(b)). */
pk_status pk_diag_gemm(int M, int N, int K, const float *A, const float *W, const float *bias, int epi,
                       const float *resid, float alpha, float *out);
/* The bf16-operand / fp32-accumulate GEMM of pk_config.gemm_bf16 (W is given in fp32 and rounded here like at upload); K % 64 == 0. */
pk_status pk_diag_gemm_bf16(int M, int N, int K, const float *A, const float *W, const float *bias, int epi,
                            const float *resid, float alpha, float *out);
/* The same with the activations stored as bf16 before the product (rounded here; in the engine the producing kernel's epilogue stores them so):
 * large shapes take the direct-to-LDS kernel (kernels/gemm_bf16_glds.hpp). */
pk_status pk_diag_gemm_bf16_a16(int M, int N, int K, const float *A, const float *W, const float *bias, int epi,
                                const float *resid, float alpha, float *out);
/* The small-M form of that product with the LayerNorm of its input rows folded in (kernels/gemm_smallm_bf16.hip; the streaming chunks of the
 * tolerance-class mode): out = epi(bf16(LayerNorm(A; gamma, beta, eps)) * bf16(W)^T + bias).  M <= 128, K = 256 * (1 .. 8; glu: .. 4).
 * PK_ERR_UNSUPPORTED for any other shape. */
/* The bf16 diag products (pk_diag_gemm_bf16*, pk_diag_ln_gemm_bf16, pk_diag_glu_dwconv_bf16, pk_diag_ffn_bf16_smallm) hand the small-M kernel
 * its weights ALSO as operand tiles (kernels.hpp GemmArgs::W_t16), as a streaming session does; on = 0 keeps the natural layout only.  Process-wide
 * test switch; both give the same bits. */
pk_status pk_diag_smallm_bf16_tiles(int on);
pk_status pk_diag_ln_gemm_bf16(int M, int N, int K, const float *A, const float *gamma, const float *beta, float eps, const float *W,
                               const float *bias, int epi, const float *resid, float alpha, float *out);
/* Two LayerNorms in front of a product (a block's final_norm_ folded, with the next block's first norm, into that block's fc1; streaming,
 * tolerance-class mode): out = silu(bf16(LN(LN(A; pre_gamma, pre_beta); gamma, beta)) bf16(W)^T + bias), pre_out [M][K] = LN(A; pre_gamma, pre_beta)
 * (the residual stream of the block that starts there).  M <= 128, K = 256 * (1 .. 8). */
pk_status pk_diag_ln2_gemm_bf16(int M, int N, int K, const float *A, const float *pre_gamma, const float *pre_beta, const float *gamma, const float *beta,
                                float eps, const float *W, const float *bias, float *out, float *pre_out);
/* The conv module's first half on a streaming chunk of the tolerance-class mode: GLU(bf16(LayerNorm(A)) bf16(W)^T + bias) -> causal depthwise
 * conv (kernel 9) over [cache_in ; the c new rows] of every stream -> BatchNorm -> SiLU (reference src/streaming_encoder.cpp:41-78).  A = [n_streams * c][d]
 * rows, stream-major; W [2 d][d]; cache_in / cache_out [n_streams][8][d]; gamma = beta = NULL: A is taken as it is.  fused = 1: the conv runs in
 * the product's epilogue (kernels.hpp DwTail; c = 1, 2 or 4), fused = 0: the separate kernel -- both bit for bit the same (tests/test_gpu_bf16.py). */
pk_status pk_diag_glu_dwconv_bf16(int n_streams, int c, int d, const float *A, const float *gamma, const float *beta, float eps, const float *W,
                                  const float *bias, const float *cache_in, int has_cache, const float *dw_w, const float *dw_bias,
                                  const float *bn_mean, const float *bn_rstd, const float *bn_g, const float *bn_b, int fused, float *out,
                                  float *cache_out);
/* One feed-forward module of a streaming chunk in the tolerance-class mode, on the small-M bf16 kernel: out = x + 0.5 * (W2 bf16(silu(W1 bf16(LN(x)) + b1)) + b2)
 * (reference src/encoder.cpp:36-46), x [M][d], W1 [f][d], W2 [d][f].  act_tiles = 1: the fc1 activations travel in the kernel's 8-row operand
 * tiles (kernels.hpp GemmArgs::out_t8 / a_t8; M % 8 == 0), 0: as rows -- bit for bit the same (tests/test_gpu_bf16.py). */
pk_status pk_diag_ffn_bf16_smallm(int M, int d, int f, const float *x, const float *gamma, const float *beta, float eps, const float *W1, const float *b1,
                                  const float *W2, const float *b2, int act_tiles, float *out);
pk_status pk_diag_layernorm(const float *x, int64_t rows, int d, const float *gamma, const float *beta, float eps, float *y);
/* LayerNorm + product on a LARGE fp32 batch (round 6; reference src/encoder.cpp:40-41, :60-61, :182-183: norm, then Linear / pointwise conv):
 * out = epi(LN(X; gamma, beta, eps) W^T + bias), X [M][K], W [N][K] (glu: [2N][K]), epi 0 none / 1 relu / 2 silu / 4 glu.
 * fold = 0: the separate LayerNorm launch, then the tile GEMM on the normalised rows.  fold = 1: a statistics pass ({mean, rstd} per row) and the tile
 * GEMM normalising while it stages its A tiles (kernels/gemm_pipe.hpp, LNA) -- what the encoder of a batch runs; bit for bit the same.
 * pre_gamma / pre_beta (both or neither): X = LN(A; pre_gamma, pre_beta) first -- a block's final_norm_ in front of the next block's ffn1 norm; y1 (optional,
 * [M][K]) receives X.  fold = 1 then writes X and takes the statistics of X's rows in one launch (launch_layernorm_then_stats).
 * PK_ERR_UNSUPPORTED where the engine would not fold (M <= 1536, N < 1024 for the non-glu epilogues, K % 32). */
pk_status pk_diag_ln_gemm(int M, int N, int K, const float *A, const float *pre_gamma, const float *pre_beta, const float *gamma, const float *beta, float eps,
                          const float *W, const float *bias, int epi, int fold, float *out, float *y1);
/* sum64 of each row of x[rows][n] (the canonical wavefront reduction). */
pk_status pk_diag_sum64(const float *x, int rows, int n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* PARAKEET_AMD_H */
