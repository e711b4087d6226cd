"""parakeet.cpp_amd -- MI355X-native (gfx950) Parakeet ASR hot path.

Python here is plumbing for tests and bench.py: configuration presets, the
seeded synthetic-input generator, the hipcc build driver and a ctypes binding
of the C ABI declared in include/parakeet_amd.h.  The product is the shared
library csrc/ builds (libparakeet_amd.so) plus the header-only C++ facade in
parakeet.cpp_amd/include/parakeet/ that mirrors the reference's
parakeet::Transcriber API.
"""
from .config import (ModelConfig, PRESETS, SortformerConfig, make_110m_config, make_eou_120m_config, make_nemotron_600m_config,  # noqa: F401
                     make_nest_encoder_config, make_rnnt_600m_config, make_sortformer_117m_config, make_tdt_600m_config, make_tiny_config)

__all__ = ["ModelConfig", "PRESETS", "make_110m_config", "make_tdt_600m_config",
           "make_rnnt_600m_config", "make_nemotron_600m_config", "make_eou_120m_config", "make_tiny_config",
           "SortformerConfig", "make_sortformer_117m_config", "make_nest_encoder_config"]
