// parakeet/config.hpp -- model configuration records of the drop-in C++ facade.
//
// Source-compatible with the reference's include/parakeet/config.hpp:9-135 (same struct, field and factory
// names, same preset values) so user code that fills or reads these compiles unchanged.  Internally every
// record flattens to the C ABI's pk_config (include/parakeet_amd.h) through detail::flatten().
#pragma once

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../../include/parakeet_amd.h"

namespace parakeet {

struct EncoderConfig {              // FastConformer encoder
    int mel_bins = 80;
    int subsampling_factor = 8;     // informational (the conv stack is fixed at 8x, as in the reference)
    int subsampling_channels = 256;
    int hidden_size = 1024;
    int num_layers = 24;
    int num_heads = 8;
    int ffn_intermediate = 4096;
    int conv_kernel_size = 9;
    float dropout = 0.1f;           // inference: identity
    float layer_norm_eps = 1e-5f;   // informational: like the reference, LayerNorm uses its default 1e-5
};

struct CTCConfig {
    EncoderConfig encoder;
    int vocab_size = 1025;          // 1024 pieces + blank
};

struct PredictionConfig {           // RNNT/TDT prediction network
    int vocab_size = 1025;
    int pred_hidden = 640;
    int num_lstm_layers = 2;
    float dropout = 0.1f;
};

struct JointConfig {
    int encoder_hidden = 1024;
    int pred_hidden = 640;
    int joint_hidden = 640;
    int vocab_size = 1025;
};

struct RNNTConfig {
    EncoderConfig encoder;
    PredictionConfig prediction;
    JointConfig joint;
};

struct TDTConfig {
    EncoderConfig encoder;
    PredictionConfig prediction;
    JointConfig joint;
    std::vector<int> durations = {0, 1, 2, 3, 4};
};

struct TDTCTCConfig {
    EncoderConfig encoder;
    PredictionConfig prediction;
    JointConfig joint;
    std::vector<int> durations = {0, 1, 2, 3, 4};
    int ctc_vocab_size = 1025;
};

namespace detail {

inline void set_encoder(EncoderConfig &e, int mel, int hidden, int layers, int ffn) {
    e.mel_bins = mel; e.hidden_size = hidden; e.num_layers = layers; e.ffn_intermediate = ffn;
    e.num_heads = 8; e.subsampling_channels = 256; e.conv_kernel_size = 9;
}
inline void set_decoder(PredictionConfig &p, JointConfig &j, int enc_hidden, int vocab, int lstm_layers) {
    p.vocab_size = vocab; p.pred_hidden = 640; p.num_lstm_layers = lstm_layers;
    j.encoder_hidden = enc_hidden; j.pred_hidden = 640; j.joint_hidden = 640; j.vocab_size = vocab;
}

inline pk_config flatten(const EncoderConfig &e, const PredictionConfig &p, const JointConfig &j, const std::vector<int> &durations,
                         int ctc_vocab, const char *joint_prefix, bool rnnt_head, int blank_id) {
    pk_config c;
    std::memset(&c, 0, sizeof c);
    c.mel_bins = e.mel_bins; c.subsampling_channels = e.subsampling_channels; c.hidden_size = e.hidden_size;
    c.num_layers = e.num_layers; c.num_heads = e.num_heads; c.ffn_intermediate = e.ffn_intermediate;
    c.conv_kernel_size = e.conv_kernel_size;
    c.vocab_size = j.vocab_size; c.pred_hidden = p.pred_hidden; c.num_lstm_layers = p.num_lstm_layers; c.joint_hidden = j.joint_hidden;
    c.num_durations = rnnt_head ? 0 : (int)durations.size();
    for (int i = 0; i < c.num_durations && i < 8; ++i) c.durations[i] = durations[i];
    c.ctc_vocab_size = ctc_vocab; c.blank_id = blank_id; c.max_symbols_per_step = 10;
    c.joint_pred_bias = 0; c.rnnt_head = rnnt_head ? 1 : 0;
    c.stft_window_centered = 0;   // switch A1, see include/parakeet_amd.h
    c.gemm_bf16 = 0;              // fp32 chains; 1 = bf16 operands / fp32 accumulate (include/parakeet_amd.h)
    std::snprintf(c.joint_prefix, sizeof c.joint_prefix, "%s", joint_prefix);
    return c;
}

}  // namespace detail

// nvidia/parakeet-tdt_ctc-110m
inline TDTCTCConfig make_110m_config() {
    TDTCTCConfig cfg;
    detail::set_encoder(cfg.encoder, 80, 512, 17, 2048);
    detail::set_decoder(cfg.prediction, cfg.joint, 512, 1025, 1);
    cfg.ctc_vocab_size = 1025;
    return cfg;
}

// nvidia/parakeet-tdt-0.6b-v3 (multilingual, 128 mel bins, 8192 pieces + blank)
inline TDTConfig make_tdt_600m_config() {
    TDTConfig cfg;
    detail::set_encoder(cfg.encoder, 128, 1024, 24, 4096);
    detail::set_decoder(cfg.prediction, cfg.joint, 1024, 8193, 2);
    return cfg;
}

// nvidia/parakeet-rnnt-0.6b
inline RNNTConfig make_rnnt_600m_config() {
    RNNTConfig cfg;
    detail::set_encoder(cfg.encoder, 80, 1024, 24, 4096);
    detail::set_decoder(cfg.prediction, cfg.joint, 1024, 1025, 2);
    return cfg;
}

}  // namespace parakeet
