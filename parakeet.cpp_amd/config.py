"""Model configuration presets, field-for-field with the reference's
include/parakeet/config.hpp:9-135 (EncoderConfig, PredictionConfig, JointConfig,
TDTCTCConfig / TDTConfig / RNNTConfig and make_110m_config / make_tdt_600m_config /
make_rnnt_600m_config), flattened into one record the C ABI (pk_config) takes."""
from dataclasses import dataclass, field, replace
from typing import List


@dataclass
class ModelConfig:
    name: str = "tdt-ctc-110m"
    # EncoderConfig (config.hpp:9-20)
    mel_bins: int = 80
    subsampling_channels: int = 256
    hidden_size: int = 512
    num_layers: int = 17
    num_heads: int = 8
    ffn_intermediate: int = 2048
    conv_kernel_size: int = 9
    # PredictionConfig / JointConfig (config.hpp:31-46)
    vocab_size: int = 1025          # label vocab incl. blank
    pred_hidden: int = 640
    num_lstm_layers: int = 1
    joint_hidden: int = 640
    durations: List[int] = field(default_factory=lambda: [0, 1, 2, 3, 4])
    ctc_vocab_size: int = 1025      # 0 = no CTC head
    joint_prefix: str = "tdt_joint_."   # tdt_ctc.cpp:5-9 ; "joint_." for ParakeetTDT / ParakeetRNNT
    head: str = "tdt"               # "tdt" (label+duration heads) or "rnnt" (single out_proj_)
    # decode defaults (tdt.hpp:71-74, ctc.hpp:55-56)
    blank_id: int = 1024
    max_symbols_per_step: int = 10
    # switch A1 (pk_config.stft_window_centered): False = Hann window left-aligned in the FFT frame, as the reference
    # author's own feature check does (scripts/compare_features.py:33-37); True = centred like torch.stft / NeMo
    stft_window_centered: bool = False
    # pk_config.gemm_bf16: encoder-side products on bf16 operands / fp32 accumulation (BASELINE configs[2] precision)
    gemm_bf16: bool = False
    # encoder-only uses (Sortformer's NEST encoder): vocab_size = 0 -> no prediction net / joint
    xscaling: bool = False          # StreamingEncoderConfig::xscaling (streaming_encoder.cpp:402-406)
    mel_normalize: bool = True      # AudioConfig::normalize (audio.cpp:140); Sortformer runs on the raw log-mel (main.cpp:516)
    encoder_prefix: str = "encoder_."   # module name of the FastConformer in the state dict ("nest_encoder_." in Sortformer)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


def make_110m_config() -> ModelConfig:           # config.hpp:77-95
    return ModelConfig()


def make_tdt_600m_config() -> ModelConfig:       # config.hpp:98-116 ; blank = vocab-1 (main.cpp:252)
    return ModelConfig(name="tdt-600m", mel_bins=128, hidden_size=1024, num_layers=24, num_heads=8,
                       ffn_intermediate=4096, vocab_size=8193, num_lstm_layers=2, ctc_vocab_size=0,
                       joint_prefix="joint_.", blank_id=8192)


def make_rnnt_600m_config() -> ModelConfig:      # config.hpp:119-135
    return ModelConfig(name="rnnt-600m", mel_bins=80, hidden_size=1024, num_layers=24, num_heads=8,
                       ffn_intermediate=4096, vocab_size=1025, num_lstm_layers=2, ctc_vocab_size=0,
                       joint_prefix="joint_.", head="rnnt", durations=[], blank_id=1024)


def make_nemotron_600m_config() -> ModelConfig:  # nemotron.hpp:31-52 (streaming; att_context_left 70, right = latency_frames)
    # blank_id: NemotronTranscriber::transcribe_chunk calls rnnt_streaming_decode_chunk with its DEFAULT blank_id = 1024
    # (nemotron.cpp:40-42, eou.hpp:91-94) although the vocabulary has 8193 entries -- kept literally.
    return ModelConfig(name="nemotron-600m", mel_bins=80, hidden_size=1024, num_layers=24, num_heads=8, ffn_intermediate=4096,
                       vocab_size=8193, num_lstm_layers=2, ctc_vocab_size=0, joint_prefix="joint_.", blank_id=1024)


def make_eou_120m_config() -> ModelConfig:       # eou.hpp:34-56 (streaming; att_context 70 / 1); ParakeetEOU registers no CTC module
    return ModelConfig(name="eou-120m", ctc_vocab_size=0, joint_prefix="joint_.")


@dataclass
class SortformerConfig:              # include/parakeet/sortformer.hpp:28-41
    nest_encoder: ModelConfig = None
    transformer_hidden: int = 192
    transformer_layers: int = 18
    transformer_heads: int = 8
    transformer_ffn: int = 768
    pre_ln: bool = False             # NeMo sortformer: post-norm
    has_final_norm: bool = False
    max_speakers: int = 4
    activity_threshold: float = 0.5
    att_context_left: int = 70       # nest_encoder.att_context_left / right (sortformer.hpp:53-54): diarize_chunk
    att_context_right: int = 0


def make_nest_encoder_config(**kw) -> ModelConfig:
    """The encoder half of make_sortformer_117m_config (sortformer.hpp:45-58): 17-layer FastConformer, 128 mels, xscaling."""
    cfg = ModelConfig(name="sortformer-nest", mel_bins=128, hidden_size=512, num_layers=17, num_heads=8, ffn_intermediate=2048,
                      vocab_size=0, ctc_vocab_size=0, durations=[], xscaling=True, mel_normalize=False, encoder_prefix="nest_encoder_.")
    return replace(cfg, **kw)


def make_sortformer_117m_config() -> SortformerConfig:     # sortformer.hpp:43-76
    return SortformerConfig(nest_encoder=make_nest_encoder_config())


def make_tiny_config(**kw) -> ModelConfig:
    """Small model of the same architecture for fast parity tests (not a reference preset)."""
    cfg = ModelConfig(name="tiny", mel_bins=80, subsampling_channels=32, hidden_size=128, num_layers=2,
                      num_heads=2, ffn_intermediate=256, vocab_size=65, pred_hidden=64, num_lstm_layers=1,
                      joint_hidden=64, ctc_vocab_size=65, blank_id=64)
    return replace(cfg, **kw)


PRESETS = {
    "tdt-ctc-110m": make_110m_config,
    "tdt-600m": make_tdt_600m_config,
    "rnnt-600m": make_rnnt_600m_config,
    "nemotron-600m": make_nemotron_600m_config,
    "eou-120m": make_eou_120m_config,
    "tiny": make_tiny_config,
}
