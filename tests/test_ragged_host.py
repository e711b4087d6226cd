"""CPU-side checks of the mixed-length (ragged) batching's HOST logic -- no GPU, no compute: the packing policy of the one-call API
(pk_plan_batches; the reference's roadmap item "batch inference: pad + length-mask", README.md:513, done by packing) and the per-clip
extents the ragged kernels' tables are built from (pk_ragged_extents), against the reference's own length formulas
(preprocess_audio n_frames = 1 + n / hop, src/audio.cpp:100-158; ConvSubsampling three stride-2 stages, src/encoder.cpp:208-217)."""
import numpy as np
import pytest

from parakeet_cpp_amd import capi


def sub_len(n):
    for _ in range(3):
        n = (n - 1) // 2 + 1
    return n


def rows_of(n):
    return sub_len(1 + int(n) // 160)


def test_plan_sorts_longest_first_and_respects_both_caps():
    """<= 256 clips and <= 8192 ENCODER ROWS per batch (the row count at which the encoder's GEMM tile grids fill whole rounds of the 256 CUs)."""
    rng = np.random.default_rng(0)
    lens = rng.integers(300, 480001, 1000)
    rows = np.array([rows_of(n) for n in lens])
    b, p, nb = capi.plan_batches(lens)
    assert nb == b.max() + 1 and nb >= 2
    longest_of = [lens[b == k].max() for k in range(nb)]
    shortest_of = [lens[b == k].min() for k in range(nb)]
    for k in range(nb):
        idx = np.where(b == k)[0]
        assert len(idx) <= 256
        assert rows[idx].sum() <= 8192 or len(idx) == 1
        assert sorted(p[idx].tolist()) == list(range(len(idx))), "positions inside a batch are 0 .. n-1"
        inside = idx[np.argsort(p[idx])]
        assert (np.diff(lens[inside]) <= 0).all(), "longest first inside a batch"
        if k:
            assert longest_of[k] <= shortest_of[k - 1], "batches are runs of the length-sorted order"
    # a batch is closed only because the next clip would not fit (or 256 clips are in it)
    order = np.lexsort((np.arange(len(lens)), -lens))
    done = 0
    for k in range(nb - 1):
        done += (b == k).sum()
        nxt = rows[order[done]]
        assert (b == k).sum() == 256 or rows[b == k].sum() + nxt > 8192


def test_plan_edge_cases():
    b, p, nb = capi.plan_batches([160000] * 65)
    assert nb == 1 and (b == 0).all() and p.tolist() == list(range(65)), "65 x 10 s = 8190 rows is exactly one batch (stable order)"
    b, p, nb = capi.plan_batches([160000] * 66)
    assert nb == 2 and (b == 0).sum() == 65
    b, p, nb = capi.plan_batches([480000] * 22)
    assert nb == 2 and (b == 0).sum() == 21, "21 x 30 s = 7896 rows; the 22nd clip would start another round of GEMM tiles"
    b, p, nb = capi.plan_batches([8193 * 1280, 300])
    assert nb == 2 and b.tolist() == [0, 1], "a clip of more rows than the budget travels alone"
    b, p, nb = capi.plan_batches([300] * 600)
    assert nb == 3 and [(b == k).sum() for k in range(3)] == [256, 256, 88], "the clip cap closes batches of short clips"
    b, p, nb = capi.plan_batches([257])
    assert nb == 1
    with pytest.raises(capi.PkError):
        capi.plan_batches([16000, 256])          # one STFT frame with reflect padding needs more than n_fft / 2 samples


def test_extents_follow_the_reference_formulas():
    lens = np.array([257, 400, 1599, 1600, 1601, 16000, 160000, 479999, 480000, 31999], np.int64)
    tm, t, tot = capi.ragged_extents(lens)
    assert tm.tolist() == [1 + n // 160 for n in lens]
    assert t.tolist() == [sub_len(1 + n // 160) for n in lens]
    L = capi.lib()
    assert all(L.pk_mel_num_frames(int(n)) == a for n, a in zip(lens, tm)) and all(L.pk_encoder_num_frames(int(a)) == c for a, c in zip(tm, t))
    assert tot[0] == lens.sum() and tot[1] == tm.sum() and tot[3] == t.sum()
    h2 = [((a - 1) // 2 + 1 - 1) // 2 + 1 for a in tm]
    assert tot[2] == sum(h2)
    assert tot[4] == sum(-(-x // 32) for x in t), "one attention unit per 32 query rows of ONE utterance (blocks never straddle clips)"
    dw = 2 if t.sum() <= 2048 else 8
    assert tot[5] == sum(-(-x // dw) for x in t)
    c1 = 2 if sum(h2) <= 1024 else 8
    assert tot[6] == sum(-(-x // c1) for x in h2)
