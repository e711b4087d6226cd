#!/usr/bin/env python3
"""Generate tests/golden/ref_compare_encoder_110m_seed42.npz by RUNNING the reference author's own PyTorch restatement
of the encoder, /root/reference/scripts/compare_encoder.py, on this repo's seeded synthetic tdt-ctc-110m weights.

Why this script: the reference's C++ arithmetic lives in the un-vendored `axiom` submodule and cannot be built here
(SURVEY.md section 8c), but scripts/compare_encoder.py is the reference author's layer-by-layer PyTorch model of the same
network, loading the same safetensors names.  It is executed in place (read from /root/reference at generation time, never
copied into this repo) inside a scratch directory that provides the files it opens:
    models/model.safetensors        <- parakeet_cpp_amd.synth.synth_weights(make_110m_config(), seed=42)
    models/debug_features_py.npy    <- seeded N(0,1) features, shape (1, 744, 80)   (the script hard-codes T = 93)
    models/vocab.txt                <- parakeet_cpp_amd.synth.synth_vocab(1024) (the script detokenises at the end)
    models/debug_*.bin              <- empty files (its C++-dump comparisons then just print a shape mismatch)
What is pinned: ConformerBlock (FeedForward, rel-pos attention incl. rel_shift, conv module, final norm) x 17, the CTC head,
log-softmax argmax and greedy collapse.  What is NOT: the script's ConvSubsampling applies SiLU where the C++ applies ReLU
(src/encoder.cpp:224,228,232 -- the code wins), so its `sub_out` is stored only as the INPUT of the block stack.
/root/reference does not exist on the GPU box: tests only read the committed .npz.
usage (in the build container): python tools/make_golden_from_reference.py"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
REF_SCRIPT = "/root/reference/scripts/compare_encoder.py"
OUT = os.path.join(ROOT, "tests", "golden", "ref_compare_encoder_110m_seed42.npz")
FEATURE_SEED = 20260926


def main():
    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import synth
    cfg = pk.make_110m_config()
    W = synth.synth_weights(cfg, seed=42)
    feats = np.random.default_rng(FEATURE_SEED).standard_normal((1, 744, 80)).astype(np.float32)
    src = open(REF_SCRIPT).read()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "models"))
        synth.save_weights(os.path.join(td, "models", "model.safetensors"), W)
        np.save(os.path.join(td, "models", "debug_features_py.npy"), feats)
        synth.save_vocab(os.path.join(td, "models", "vocab.txt"), synth.synth_vocab(cfg.vocab_size - 1))
        for n in ("after_conv1", "after_block1", "after_block2", "subsampling_out", "after_layer0"):
            open(os.path.join(td, "models", f"debug_{n}.bin"), "wb").close()
        ns = {"__name__": "__ref_compare_encoder__"}
        os.chdir(td)
        try:
            log = io.StringIO()
            with contextlib.redirect_stdout(log):
                exec(compile(src, REF_SCRIPT, "exec"), ns)
        finally:
            os.chdir(cwd)
    g = lambda k: ns[k].detach().numpy().astype(np.float32)
    sub_out, layer0, final, logp = g("sub_out"), g("layer0_out"), g("x"), g("log_probs")
    preds = ns["preds"].numpy().astype(np.int32)
    tokens = np.asarray(ns["tokens"], np.int32)
    top2 = np.sort(logp[0], axis=-1)[:, -2:]
    np.savez_compressed(OUT, feature_seed=np.int64(FEATURE_SEED), weight_seed=np.int64(42), block_input=sub_out, layer0_out=layer0,
                        encoder_out=final, ctc_argmax=preds, ctc_best_logp=logp.max(-1).astype(np.float32), ctc_tokens=tokens,
                        ctc_min_margin=np.float32((top2[:, 1] - top2[:, 0]).min()))
    print(f"wrote {OUT}: block_input {sub_out.shape}, encoder_out {final.shape}, {len(tokens)} CTC tokens, "
          f"min top-1/top-2 margin {float((top2[:, 1] - top2[:, 0]).min()):.3e}")
    print("--- tail of the reference script's own output ---")
    print("\n".join(log.getvalue().splitlines()[-8:]))


if __name__ == "__main__":
    main()
