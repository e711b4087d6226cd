/*
 * oracle/pk_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * See pk_oracle.c for the parity status and the reference citations.
 */
#ifndef PK_ORACLE_H
#define PK_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif


/* include/parakeet/audio.hpp:7-17 (AudioConfig) + switches A1/A2 */
typedef struct {
    int sample_rate;     /* 16000 */
    int n_fft;           /* 512 */
    int win_length;      /* 400 */
    int hop_length;      /* 160 */
    int n_mels;          /* 80 / 128 */
    float f_min;         /* 0 */
    float f_max;         /* <=0 -> sample_rate/2 */
    int normalize;       /* 1 */
    int window_centered; /* A1: 0 = left-aligned, as scripts/compare_features.py:33-37 checks the C++ (default of oracle.py and of
                            the product); 1 = torch.stft / NeMo placement */
    int power_via_abs;   /* A2: 1 = abs() then square (literal reference, default), 0 = re^2+im^2 */
} orc_audio_config;

/* include/parakeet/config.hpp:9-95 flattened + switches A3/A5 */
typedef struct {
    int mel_bins, sub_channels, d_model, n_layers, n_heads, ffn, conv_k;
    int vocab;        /* joint label vocab incl. blank */
    int pred_hidden, lstm_layers, joint_hidden;
    int n_durations;
    int durations[8];
    int blank_id;     /* tdt.hpp:71-74 default 1024 */
    int max_symbols;  /* 10 */
    float ln_eps;     /* A3: 1e-5 */
    float bn_eps;     /* A3: 1e-5 */
    int joint_pred_bias; /* A5: 0 = drop (literal reference), 1 = add (NeMo) */
    int gemm_bf16;    /* 0: fp32 fma chains.  1: the operands of every Linear / 1x1-conv product of the encoder, the CTC head and
                         enc_proj are rounded to bf16 first (pk_config.gemm_bf16 of the product); accumulation stays fp32 */
    char joint_prefix[32]; /* "tdt_joint_." (110M hybrid) or "joint_." (TDT/RNNT 600M) */
} orc_config;

typedef struct orc_model orc_model;

const char *orc_last_error(void);
void orc_set_threads(int n);
int orc_get_max_threads(void);

void orc_math_v(int fn, const float *in, float *out, int64_t n);
float orc_sum64_f(const float *x, int64_t n);
void orc_linear(int M, int N, int K, const float *A, const float *W, const float *bias, float *out);
void orc_linear_scalar(int M, int N, int K, const float *A, const float *W, const float *bias, float *out);
void orc_layer_norm(const float *x, int64_t rows, int d, const float *g, const float *b, float eps, float *y);

orc_model *orc_model_new(const orc_config *cfg);
void orc_model_free(orc_model *m);
int orc_model_add(orc_model *m, const char *name, const float *data, int ndim, const int64_t *shape);

void orc_mel_filterbank(int n_freqs, int n_mels, float sample_rate, float f_min, float f_max, float *fb);
int orc_mel_num_frames(int64_t n_samples, int hop);
/* src/transformer.cpp:15-88 on x[B][T][d] in place; tensors named <prefix>layers_.<i>. ... in the model */
int orc_transformer_encoder(orc_model *m, const char *prefix, int n_layers, int n_heads, int pre_ln, int has_final_norm,
                            float ln_eps, float *x, int B, int T, int d);
int orc_mel(const orc_audio_config *ac, const float *pcm, int64_t n, float *out, float *logmel_tap);
void orc_pos_emb(int seq_len, int d_model, float *pe);
int orc_subsampled_len(int n_mel_frames);
int orc_subsampling(orc_model *m, const float *feats, int B, int Tm, float *out, float *tap_conv1, float *tap_stage3);
int orc_conformer_block(orc_model *m, int layer, float *x, int B, int T, const float *pos_emb, int stop_after);
int orc_encoder(orc_model *m, const float *feats, int B, int Tm, float *out, float *layer_taps);
int orc_ctc_logprobs(orc_model *m, const float *enc, int B, int T, float *logp);
void orc_ctc_greedy(const float *logp, int B, int T, int V, int blank_id, int32_t *ids, int32_t *lens,
                    int32_t *start, int32_t *end, float *conf);
int orc_tdt_greedy(orc_model *m, const float *enc, int B, int T, int max_tokens, int max_steps, int32_t *ids,
                   int32_t *lens, int32_t *start, int32_t *end, float *conf, int32_t *steps, float *first_label_logp);
int orc_rnnt_greedy(orc_model *m, const float *enc, int B, int T, int max_tokens, int32_t *ids, int32_t *lens,
                    int32_t *start, float *conf);

/* ---- Sortformer (src/sortformer.cpp) ---- */
void orc_model_set_encoder(orc_model *m, const char *prefix, int xscaling);
int orc_sortformer_forward(orc_model *m, const float *feats, int B, int Tm, int n_tlayers, int n_theads, int pre_ln,
                           int has_final_norm, float *probs);
int orc_probs_to_segments(const float *probs, int T, int S, float threshold, int32_t *spk, float *start, float *end);

/* ---- phrase boosting: ContextTrie + boosted greedy decoders (src/phrase_boost.cpp) ---- */
typedef struct orc_trie orc_trie;
orc_trie *orc_trie_new(void);
void orc_trie_free(orc_trie *t);
void orc_trie_insert(orc_trie *t, const int32_t *ids, int n);
int orc_trie_size(const orc_trie *t);
int orc_trie_boosted_tokens(const orc_trie *t, const int32_t *states, int n, unsigned char *flag, int V);   /* -> count */
int orc_trie_advance(const orc_trie *t, const int32_t *states, int n, int tok, int32_t *out);                /* -> count */
void orc_ctc_greedy_boosted(const float *logp, int B, int T, int V, int blank_id, const orc_trie *trie, float boost, int32_t *ids,
                            int32_t *lens, int32_t *start, int32_t *end, float *conf);
/* orc_tdt_greedy + per utterance the smallest (top-1 minus top-2) label log-prob over all of its decisions: the early warning of the
 * tolerance-class (bf16) mode -- a margin below the mode's error is a token that may flip */
int orc_tdt_greedy_margin(orc_model *m, const float *enc, int B, int T, int max_tokens, int max_steps, int32_t *ids,
                          int32_t *lens, int32_t *start, int32_t *end, float *conf, int32_t *steps, float *min_margin,
                          float *step_margin /* optional [B][step_cap]: the margin of every decision in order */,
                          int32_t *step_label /* optional [B][step_cap]: the label every decision chose, blank included */, int step_cap);
/* teacher-forced joint scores along a given (or, labels_in == NULL, the greedy) decision path of ONE utterance: see pk_oracle.c */
int orc_tdt_score(orc_model *m, const float *enc, int T, const int32_t *labels_in, const int32_t *dur_in, int n_steps, int32_t *labels_out,
                  int32_t *dur_out, float *label_lp, float *dur_lp);
int orc_tdt_greedy_boosted(orc_model *m, const float *enc, int B, int T, int max_tokens, int max_steps, const orc_trie *trie, float boost,
                           int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf, int32_t *steps);

/* ---- streaming path (StreamingAudioPreprocessor, forward_chunk, rnnt_streaming_decode_chunk) -- one object per stream ---- */
typedef struct orc_stream orc_stream;
orc_stream *orc_stream_new(orc_model *m, int att_context_left, int att_context_right);
void orc_stream_free(orc_stream *s);
int orc_stream_mel(orc_stream *s, const float *pcm, int n, float *out);
int orc_stream_encode(orc_stream *s, const float *mel, int n_frames, float *enc, int max_out);
int orc_stream_decode(orc_stream *s, const float *enc, int c, int max_tokens, int32_t *ids, int32_t *start, int32_t *end, float *conf);
/* teacher-forced joint scores of ONE chunk along a given (or, labels_in == NULL, the greedy) decision path, state carried: see pk_oracle.c */
int orc_stream_score(orc_stream *s, const float *enc, int c, const int32_t *labels_in, const int32_t *dur_in, int n_steps, int32_t *labels_out,
                     int32_t *dur_out, float *label_lp, float *dur_lp);

int orc_sortformer_chunk(orc_stream *s, const float *feats, int n_frames, int n_tlayers, int n_theads, int pre_ln, int has_final_norm,
                         float *probs, int max_out);


#ifdef __cplusplus
}
#endif

#endif /* PK_ORACLE_H */
