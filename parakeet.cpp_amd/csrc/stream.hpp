// parakeet.cpp_amd/csrc/stream.hpp -- StreamBatch: S lock-step streams with cached encoder state (see stream.cpp).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <vector>

#include "engine.hpp"

namespace pk {

class StreamBatch {
  public:
    StreamBatch(Model &m, int n_streams, int att_left, int att_right);
    ~StreamBatch();
    StreamBatch(const StreamBatch &) = delete;
    StreamBatch &operator=(const StreamBatch &) = delete;
    void reset();
    // stage entry points (host buffers); each returns the number of frames it produced per stream (0: buffered / cached)
    int mel(const float *pcm, int n_samples, float *out /*[S][n_frames][F]*/, int cap_frames);
    int encode(const float *mel_in, int n_frames, float *enc /*[S][c][d]*/, int cap_frames);
    // the same, leaving the c output frames per stream on the device (*d_enc -> [S*c][d], valid until the next call); enqueued only
    int encode_keep(const float *mel_in, int n_frames, const float **d_enc);
    void decode(const float *enc, int c, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf);
    // rnnt_streaming_decode_chunk (src/eou.cpp:17-98) of every stream along a GIVEN decision path: stream s walks n_steps[s] decisions
    // labels[s][k] / dur_idx[s][k] (arrays pitched `cap`), the joint's outputs of every step are recorded -- label_logp[s][k][V] (may be null),
    // dur_logp[s][k][D] -- and the state each stream carries into its next chunk is the one that path leaves.  n_done[s] = steps walked.
    void score(const float *enc, int c, const int32_t *labels, const int32_t *dur_idx, const int32_t *n_steps, int cap, float *label_logp,
               float *dur_logp, int32_t *n_done);
    // NemotronTranscriber::transcribe_chunk for S streams; device-resident between the stages
    void push(const float *pcm, int n_samples, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf);
    int S;

  private:
    Model &m_;
    int left_, right_;
    // StreamingAudioPreprocessor state (host)
    std::vector<float> preemph_last_;
    std::vector<std::vector<float>> overlap_;
    // device state
    DevBuf mel_cache_;          // [S][8][F] leftover mel frames (first n_mel_cache_ valid)
    int n_mel_cache_ = 0;
    struct LayerState { DevBuf k[2], v[2], conv[2]; int cur = 0, ccur = 0, n_kv = 0, has_conv = 0; };
    std::vector<std::unique_ptr<LayerState>> layers_;
    const std::vector<Model::SigW> *sig_ = nullptr;      // the model's tiled weight copies (Model::sigma_weights); empty: natural operands
    int frame_offset_ = 0;
    Workspace ws_;              // encoder workspace of the current chunk
    Workspace wd_;              // decode workspace: h / c / token persist across chunks (never re-allocated)
    int dec_cap_frames_ = 0;
    DevBuf pre_, mel_dev_, mel_all_, enc_in_, force_, score_;
    // pinned host staging of a chunk's results: the five device-to-host copies of fetch_tokens then queue back to back behind the decode and cost ONE
    // host round trip; into the caller's pageable arrays every one of them is a blocking copy of its own (~20 us each: 5 % of a 2 ms chunk)
    void *pin_tok_ = nullptr;
    size_t pin_tok_bytes_ = 0;
    void *pinned_tokens(size_t bytes);
    // ... and of a chunk's pre-emphasised samples on their way up: the upload is asynchronous, and a push goes on to enqueue the encoder without
    // waiting for it (the pageable buffer it replaces forced a synchronisation in front of the encoder's launches)
    float *pin_pcm_ = nullptr;
    size_t pin_pcm_floats_ = 0;
    int mel_impl(const float *pcm, int n_samples, float *out, int cap_frames, bool sync);
    // The launch chain of a chunk's 24 blocks as a hipGraph (steady state of a session: full caches, the chunk shape of the last capture): one graph per
    // parity of the double-buffered caches, keyed by every pointer and size the captured launches carry.  The host then submits ~250 dependent launches
    // as one.  Measured slower than the launches it replaces (stream.cpp stream_graph(), profiles/r05_stream_graph_ab.txt): an opt-in of EXPERIMENTAL builds.
    struct EncGraph { hipGraphExec_t exec = nullptr; std::vector<uintptr_t> key; };
    EncGraph enc_graph_[2];
    DevBuf x_alt_;              // second residual-stream buffer: a block's final norm folded into the next block's first product writes it (stream.cpp)
    std::map<int, std::unique_ptr<DevBuf>> pos_tables_;   // Tp -> pos_proj of every layer [L][2Tp-1][d], natural columns
    int encode_device(const float *d_mel, int n_frames);                    // -> ws_.x [S*c][d], returns c
    const float *pos_table(int Tp);
    void decode_device(const float *d_enc, int c, int max_tokens);
};

}  // namespace pk
