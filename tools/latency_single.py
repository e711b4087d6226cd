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
n_clips = int(os.environ.get("PK_LAT_CLIPS", "1"))             # clips per call (one lock-step decode)
pcms = list(synth.synth_pcm(n_clips, 160000, seed=1234))
loop = os.environ.get("PK_LAT_DECODE_LOOP")               # phases (default) | graph | persistent
if loop: gm.set_decode_loop(loop)
for dec in ("tdt", "ctc"):
    gm.transcribe_pcm(pcms, dec)
    ts = []
    for _ in range(int(os.environ.get('PK_LAT_ITERS', '100'))):
        t0 = time.perf_counter(); r = gm.transcribe_pcm(pcms, dec); ts.append((time.perf_counter() - t0) * 1e3)
    print(loop or "phases", dec, "%d x 10 s clip%s per call, end-to-end ms (median):" % (n_clips, "" if n_clips == 1 else "s"), round(statistics.median(ts), 3), "min", round(min(ts), 3), "tokens", len(r[0]["token_ids"]))
