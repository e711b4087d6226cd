"""Multi-GPU path on CPU: world_size-2 and world_size-8 gloo processes shard a clip list exactly as bench.py / the 8-GPU driver do
(parakeet_cpp_amd.shard), 'decode' their shard with a deterministic stand-in, and all-gather the results.  Checks
the partition (disjoint, complete, batch-aligned) and the reassembly order -- there is no data-path collective to
test beyond this: utterances are independent."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT, pk  # noqa: F401
from parakeet_cpp_amd.shard import gather_results, shard_by_audio, shard_indices


def test_shard_partition_properties():
    for n, world, batch in [(8192, 8, 64), (130, 2, 64), (5, 4, 64), (64, 1, 64), (1000, 3, 16)]:
        seen = []
        for r in range(world):
            idx = shard_indices(n, r, world, batch)
            assert all(0 <= i < n for i in idx)
            for k in range(0, len(idx), batch):           # every rank works in whole batches of consecutive clips
                chunk = idx[k:k + batch]
                assert chunk == list(range(chunk[0], chunk[0] + len(chunk))) and chunk[0] % batch == 0
            seen += idx
        assert sorted(seen) == list(range(n))
    assert len(shard_indices(8192, 3, 8, 64)) == 1024         # BASELINE configs[3]: 1024 clips per GPU
    assert gather_results(["a", "b"], [1, 0], 2, 1) == ["b", "a"]


def test_shard_by_audio_properties():
    """Mixed-length corpora (round 4): every clip on exactly one rank, longest first, loads within one longest clip of each other, equal
    lengths degenerate to round-robin, and the library's own in-process partition (pk_group_transcribe_pcm) follows the same rule."""
    import numpy as np
    rng = np.random.default_rng(5)
    for n, world in [(1000, 8), (37, 2), (5, 4), (64, 1), (3, 8)]:
        lens = [int(x) for x in rng.integers(300, 480000, n)]
        seen, loads = [], []
        for r in range(world):
            idx = shard_by_audio(lens, r, world)
            assert [lens[i] for i in idx] == sorted((lens[i] for i in idx), reverse=True), "longest first on every rank"
            seen += idx
            loads.append(sum(lens[i] for i in idx))
        assert sorted(seen) == list(range(n))
        assert max(loads) - min(loads) <= max(lens), "greedy longest-first: no rank is more than one clip ahead"
    assert shard_by_audio([160000] * 16, 3, 8) == [3, 11], "equal lengths: rank r takes clips r, r + world, ..."


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    import pkload; pk = pkload.load()
    from parakeet_cpp_amd.shard import shard_indices, gather_results
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = int(os.environ.get("PK_TEST_CLIPS", "300"))
    idx = shard_indices(n, rank, world, batch=64)
    local = [[i * 7 % 13, i] for i in idx]                    # stand-in for per-clip token ids
    merged = gather_results(local, idx, n, world, dist)
    assert merged == [[i * 7 % 13, i] for i in range(n)], "reassembly order"
    import numpy as np
    from parakeet_cpp_amd.shard import gather_token_matrix
    lens = np.array([1 + i % 5 for i in idx], np.int32)
    ids = np.zeros((len(idx), 6), np.int32)
    for r, i in enumerate(idx):
        ids[r, :lens[r]] = np.arange(lens[r]) + 10 * i
    gi, gl = gather_token_matrix(ids, lens, idx, n, world, dist)          # fixed-stride all_gather_into_tensor (RCCL on the GPU box)
    assert gl.tolist() == [1 + i % 5 for i in range(n)]
    assert all(gi[i, :gl[i]].tolist() == (np.arange(gl[i]) + 10 * i).tolist() for i in range(n)), "token matrix order"
    # mixed-length corpus: the partition by audio (every rank computes it alone, no communication) and the same exchange
    from parakeet_cpp_amd.shard import shard_by_audio
    clip_len = [300 + (i * 7919) % 400000 for i in range(n)]
    idx2 = shard_by_audio(clip_len, rank, world)
    lens2 = np.array([1 + i % 5 for i in idx2], np.int32)
    ids2 = np.zeros((len(idx2), 6), np.int32)
    for r, i in enumerate(idx2):
        ids2[r, :lens2[r]] = np.arange(lens2[r]) + 10 * i
    gi2, gl2 = gather_token_matrix(ids2, lens2, idx2, n, world, dist)
    assert gl2.tolist() == [1 + i % 5 for i in range(n)] and np.array_equal(gi2, gi), "mixed-length shards reassemble to the same matrix"
    # weight distribution: rank 0 reads the file once, one broadcast, every rank builds its model from the memory image
    from parakeet_cpp_amd.shard import broadcast_file
    from parakeet_cpp_amd import capi, synth
    cfg = pk.make_tiny_config()
    wp = os.path.join({tmp!r}, "tiny.safetensors")
    if rank == 0:
        synth.save_weights(wp, synth.synth_weights(cfg, seed=42))
    dist.barrier()
    img = broadcast_file(wp if rank == 0 else "/nonexistent", rank, world, dist)
    m = capi.Model(img, cfg)                                  # pk_model_load_buffer (host side only: no GPU in this test)
    import ctypes as C
    out = capi.PkConfig()
    assert capi.lib().pk_model_config(m._h, C.byref(out)) == 0 and out.hidden_size == cfg.hidden_size
    try:
        capi.Model(img[: len(img) // 2], cfg)
        raise SystemExit("a truncated image must be refused")
    except RuntimeError as e:
        assert "safetensors" in str(e)
    import torch
    t = torch.tensor([float(len(idx))]); dist.all_reduce(t)
    assert int(t.item()) == n
    # the wall-clock reduction of bench.py / tools/transcribe_sharded.py: every rank ends with the SLOWEST rank's time
    w = torch.tensor([1.0 + rank], dtype=torch.float64); dist.all_reduce(w, op=dist.ReduceOp.MAX)
    assert w.item() == float(world)
    dist.barrier()
    if rank == 0: print("GLOO_OK", len(idx))
    dist.destroy_process_group()
""")


def _run_gloo_worker(tmp_path, world, n_clips):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, tmp=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PK_TEST_CLIPS=str(n_clips), OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def test_two_process_gloo_shard_and_gather(tmp_path):
    assert "GLOO_OK 172" in _run_gloo_worker(tmp_path, 2, 300)            # rank 0 owns batches 0, 2, 4 -> 64 + 64 + 44 clips


def test_eight_process_gloo_configs3_arithmetic(tmp_path):
    """The exact rank count and clip count of BASELINE configs[3] (8192 x 10 s clips over 8 ranks; round-5 verdict, item 7: the 8-rank arithmetic had
    only run at world size 1 and 2): partition into 1024 clips = 16 whole batches per rank, the object gather, the fixed-stride all-gather of the
    [clips][2 + max_tokens] token matrix for uniform and mixed-length shards, the weight-image broadcast, the sum- and max-reductions."""
    assert "GLOO_OK 1024" in _run_gloo_worker(tmp_path, 8, 8192)


def test_bench_spawns_eight_ranks(tmp_path):
    """`bench.py --gpus 8 --rendezvous-only`: the launcher, the barrier, the per-rank gather and the max-reduce at the world size the driver's SCALE
    run uses (no devices, no throughput claim)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--rendezvous-only"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["collective_ranks"] == 8 and line["dry_run"] is True
    assert len(line["ms_per_step_per_rank"]) == 8 and line["ms_per_step"] == max(line["ms_per_step_per_rank"])


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher must become one (round-2 verdict: run bare it silently measured one GPU).  On CPU the
    ranks meet over gloo (--rendezvous-only: launcher, barrier, per-rank gather and max-reduce only; no throughput claim)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--rendezvous-only"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["collective_ranks"] == 2 and line["dry_run"] is True
    assert len(line["ms_per_step_per_rank"]) == 2 and line["ms_per_step"] == max(line["ms_per_step_per_rank"])


def test_bench_refuses_more_gpus_than_devices():
    """A 2-GPU figure must never come from fewer devices: with no (or one) device visible `--gpus 2` fails loudly before anything is timed."""
    from parakeet_cpp_amd import capi
    if capi.device_count() >= 2:
        import pytest
        pytest.skip("this host has >= 2 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "device(s) visible" in (out.stdout + out.stderr)
    # a launcher that started a different number of ranks than --gpus says is refused too
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env2, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "must agree" in (out.stdout + out.stderr)


def test_shard_sessions_partition():
    """Streaming sessions (configs[4]): dealt to the ranks in lock-step groups; every session on exactly one rank, groups intact."""
    from parakeet_cpp_amd import shard
    for n, world, group in [(128, 8, 16), (100, 8, 16), (16, 8, 16), (33, 2, 16), (7, 3, 4)]:
        owned = [shard.shard_sessions(n, r, world, group) for r in range(world)]
        flat = sorted(i for o in owned for i in o)
        assert flat == list(range(n))
        for o in owned:
            for k in range(0, len(o), group):
                blk = o[k:k + group]
                assert blk == list(range(blk[0], blk[0] + len(blk))) and blk[0] % group == 0      # whole groups, in order
        assert max(map(len, owned)) - min(map(len, owned)) <= group


def test_bench_stream_refuses_more_gpus_than_devices():
    from parakeet_cpp_amd import capi
    if capi.device_count() >= 2:
        import pytest
        pytest.skip("this host has >= 2 devices")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_stream.py"), "--gpus", "2", "--chunks", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "device(s) visible" in (out.stdout + out.stderr)
