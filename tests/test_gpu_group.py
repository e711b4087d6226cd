"""pk_group -- the library's multi-GPU entry: one replica per device, utterance batches dealt round-robin, the weight
image broadcast and the token matrix all-gathered over RCCL.  The GPU box of the test run exposes ONE device, so the communicator has
one rank -- the collectives (ncclBroadcast, ncclAllReduce(max), ncclAllGather) still execute, which is exactly the point: the same
code path runs unchanged on 8 devices.  Results must equal pk_transcribe_pcm on a single model, clip for clip."""
import numpy as np
import pytest

import gpu_common as G
from parakeet_cpp_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_group_matches_single_model(tmp_path):
    cfg = G.tiny(name="tiny-group", vocab_size=1025, ctc_vocab_size=1025, blank_id=1024)
    wp, vp = str(tmp_path / "g.safetensors"), str(tmp_path / "g_vocab.txt")
    synth.save_weights(wp, synth.synth_weights(cfg, seed=3))
    synth.save_vocab(vp, synth.synth_vocab(cfg.vocab_size - 1))
    # 70 clips of three lengths: several batches (one of them > 64 clips -> split), ragged order
    rng = np.random.default_rng(0)
    lens = [32000] * 66 + [20011] * 3 + [48000]
    rng.shuffle(lens)
    clips = [synth.synth_pcm(1, n, seed=100 + i)[0] for i, n in enumerate(lens)]
    single = capi.Model(wp, cfg, vocab_path=vp, device=0)
    grp = capi.Group(wp, cfg, vocab_path=vp)              # every visible device
    assert grp.size() == capi.device_count() >= 1
    for dec, ts in (("tdt", True), ("ctc", False)):
        want = single.transcribe_pcm(clips, decoder=dec, timestamps=ts)
        got = grp.transcribe_pcm(clips, decoder=dec, timestamps=ts)
        assert len(got) == len(want) == 70
        for i, (g, w) in enumerate(zip(got, want)):
            assert g["token_ids"] == w["token_ids"] and g["text"] == w["text"], i
            if ts:
                assert g["start"] == w["start"] and g["end"] == w["end"] and g["conf"] == w["conf"] and g["words"] == w["words"]
        st = grp.last_stats()
        assert sum(st["clips_per_rank"]) == 70
        assert abs(st["audio_seconds"] - sum(lens) / 16000.0) < 1e-6
        assert st["wall_ms_max"] > 0
    assert sum(len(w["token_ids"]) for w in want) > 0
    grp.close(); single.close()


def test_group_explicit_device_list_and_errors(tmp_path):
    cfg = G.tiny(name="tiny-group2")
    wp = str(tmp_path / "g2.safetensors")
    synth.save_weights(wp, synth.synth_weights(cfg, seed=4))
    grp = capi.Group(wp, cfg, devices=[0])
    assert grp.size() == 1
    out = grp.transcribe_pcm([synth.synth_pcm(1, 16000, seed=1)[0]], decoder="tdt")
    assert len(out) == 1
    grp.close()
    with pytest.raises(RuntimeError, match="out of range"):
        capi.Group(wp, cfg, devices=[99])
    with pytest.raises(RuntimeError, match="Cannot open"):
        capi.Group(str(tmp_path / "nope.safetensors"), cfg)
