#!/usr/bin/env python3
"""The reference's OWN benchmark protocol on the MI355X engine, for a like-for-like line next to its published numbers
(reference src/bench.cpp:149,180-192, README.md:447-466): encoder forward only, batch 1, random features
`randn(1, seconds * 100, mel_bins)`, 1 warm-up, then timed iterations; audio lengths 1 / 5 / 10 / 30 / 60 s.
Timed per call: pk_encode on host buffers, i.e. H2D of the features + subsampling + all conformer blocks + D2H of the encoder output
(the reference times model.encoder()(features) on device tensors).  Prints one JSON object; profiles/r01_reference_protocol.json."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PUBLISHED_M3 = {   # README.md:453-466 of the reference: (CPU ms, GPU ms) on an Apple M3
    "tdt-ctc-110m": {1: (262, 24), 5: (1222, 26), 10: (2581, 27), 30: (10061, 32), 60: (26559, 72)},
    "tdt-600m": {10: (10779, 520)},
    "rnnt-600m": {10: (10648, 1468)},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="tdt-ctc-110m,tdt-600m,rnnt-600m")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--seconds", default="", help="comma-separated audio lengths instead of the reference's table (profiling one shape)")
    args = ap.parse_args()
    import numpy as np
    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import capi, synth
    out = {"protocol": "encoder forward only, batch 1, randn(1, 100 * seconds, mel) features, 1 warm-up, median of N (reference src/bench.cpp)",
           "iters": args.iters, "dtype": "f32", "device": "1x MI355X", "rows": []}
    for name in args.models.split(","):
        cfg = pk.PRESETS[name]()
        wp = f"/tmp/pk_refproto_{name}.safetensors"
        if not os.path.exists(wp):
            synth.save_weights(wp, synth.synth_weights(cfg, seed=42))
        gm = capi.Model(wp, cfg, device=0)
        secs = [1, 5, 10, 30, 60] if name == "tdt-ctc-110m" else [10]
        if args.seconds:
            secs = [int(x) for x in args.seconds.split(",")]
        for sec in secs:
            feats = np.random.default_rng(sec).standard_normal((1, sec * 100, cfg.mel_bins)).astype(np.float32)
            gm.encode(feats)                                            # 1 warm-up (also builds the position tables for this length)
            ts = []
            for _ in range(args.iters):
                t0 = time.perf_counter()
                gm.encode(feats)
                ts.append((time.perf_counter() - t0) * 1e3)
            ms = statistics.median(ts)
            pub = PUBLISHED_M3.get(name, {}).get(sec)
            out["rows"].append({"model": name, "audio_s": sec, "encoder_ms": round(ms, 3), "rtfx": round(sec * 1e3 / ms, 1),
                                "reference_m3_cpu_ms": pub[0] if pub else None, "reference_m3_gpu_ms": pub[1] if pub else None,
                                "speedup_vs_reference_gpu": round(pub[1] / ms, 1) if pub else None})
        gm.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
