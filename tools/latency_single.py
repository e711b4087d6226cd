"""End-to-end latency of Transcriber::transcribe on ONE 10 s clip (pk_transcribe_pcm: H2D, mel, encoder, decode, D2H, detokenise)."""
import sys, time, statistics
sys.path.insert(0, '.')
import numpy as np, pkload
pk = pkload.load()
from parakeet_cpp_amd import capi, synth
cfg = pk.make_110m_config()
import os
wp = "/tmp/pk_lat.safetensors"
if not os.path.exists(wp): synth.save_weights(wp, synth.synth_weights(cfg, seed=42))
gm = capi.Model(wp, cfg, device=0)
pcm = synth.synth_pcm(1, 160000, seed=1234)[0]
for dec in ("tdt", "ctc"):
    gm.transcribe_pcm([pcm], dec)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); r = gm.transcribe_pcm([pcm], dec); ts.append((time.perf_counter() - t0) * 1e3)
    print(dec, "one 10 s clip end-to-end ms (median):", round(statistics.median(ts), 3), "tokens", len(r[0]["token_ids"]))
