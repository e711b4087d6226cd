#!/usr/bin/env python3
"""tools/experiments/bf16_layout_check.py OUT.npy -- encoder output of a 3-layer cut of tdt-600m in the bf16 mode on 24 x 30 s clips (9024 rows:
the large-batch kernels).  Run under different PK_BF16_PERSIST settings of the EXPERIMENTAL library and compare the files: the blocked
fc1 -> fc2 hand-off and the register epilogue must not change a single bit against the row-major LDS-epilogue form (same products, same k order)."""
import dataclasses
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import pkload
pk = pkload.load()
from parakeet_cpp_amd import capi, synth

cfg = dataclasses.replace(pk.make_tdt_600m_config(), num_layers=3, gemm_bf16=True, name="tdt-600m-3L-bf16-layout")
W = synth.synth_weights(cfg, seed=11)
with tempfile.TemporaryDirectory() as td:
    wp = os.path.join(td, "w.safetensors")
    synth.save_weights(wp, W)
    m = capi.Model(wp, cfg, device=0)
    pcm = synth.synth_pcm(24, 480000, seed=5)
    enc = m.encode(m.mel(pcm))
    np.save(sys.argv[1], enc)
    print(sys.argv[1], enc.shape, float(np.abs(enc).max()), int(np.bitwise_xor.reduce(enc.view(np.uint32).ravel())))
    m.close()
