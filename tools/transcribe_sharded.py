#!/usr/bin/env python3
"""Utterance-sharded transcription driver -- BASELINE.json configs[3] (8192 x 10 s clips = 128 batches of 64, 1024 clips per GPU on
8 GPUs).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/transcribe_sharded.py --clips 8192

Rank r owns batches r, r+G, ... (parakeet_cpp_amd.shard.shard_indices), runs them through the resident two-stream pipeline
(pk_batch_*: decode(k) under encoder(k+1)); there is NO data-path collective.  After the last batch the token ids travel in one
fixed-stride all_gather (RCCL over xGMI; ~2 MB) and the wall time in one max-all-reduce.  Prints one JSON line on rank 0.
Synthetic clips are a pure function of the clip's batch index, so any world size transcribes the same 'corpus' and
`--digest` values are comparable across runs (the reference's e2e invariant: same ids whatever the sharding)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n_clips, batch, clip_samples, decoder, cfg_name="tdt-ctc-110m", layers=None, dist=None, rank=0, world=1, local_rank=0, barrier=None,
        broadcast_weights=False):
    import dataclasses
    import numpy as np
    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import capi, shard, synth
    cfg = pk.PRESETS[cfg_name]()
    if layers:
        cfg = dataclasses.replace(cfg, num_layers=layers, name=f"{cfg.name}-{layers}L")
    wpath = f"/tmp/pk_sharded_{cfg.name}_{cfg.num_layers}_seed42.safetensors"
    if local_rank == 0 and not os.path.exists(wpath):
        synth.save_weights(wpath + f".{os.getpid()}", synth.synth_weights(cfg, seed=42))
        os.replace(wpath + f".{os.getpid()}", wpath)
    if barrier:
        barrier()
    if broadcast_weights:      # rank 0 reads the file once; one RCCL broadcast; replicas are built from memory
        img = shard.broadcast_file(wpath, rank, world, dist, device=f"cuda:{local_rank}" if world > 1 else None)
        model = capi.Model(img, cfg, device=local_rank)
    else:
        model = capi.Model(wpath, cfg, device=local_rank)
    idx = shard.shard_indices(n_clips, rank, world, batch)
    bt = capi.Batch(model, batch, clip_samples)
    mt = None
    ids_l, lens_l = [], []
    chunks = [idx[k:k + batch] for k in range(0, len(idx), batch)]
    POOL = 4                                               # distinct synthetic batches, generated before the clock starts
    pool = [synth.synth_pcm(batch, clip_samples, seed=1234 + g) for g in range(POOL)]
    gen = lambda c: pool[(c[0] // batch) % POOL][:len(c)]                                     # function of the GLOBAL batch index
    if barrier:
        barrier()
    t0 = time.perf_counter()
    # producer / consumer around the two-stream pipeline: run(k) enqueues encoder(k) and drives decode(k-1) on the host thread;
    # while encoder(k) is still running the host fetches results(k-1) (no flush) and stages PCM(k+1) on the copy stream
    if chunks:
        bt.upload_async(gen(chunks[0]))
    for k, c in enumerate(chunks):
        bt.run(decoder)
        if k >= 1:
            r = bt.results_done()
            ids_l.append(r["ids"]); lens_l.append(r["lens"])
        if k + 1 < len(chunks):
            bt.upload_async(gen(chunks[k + 1]))
    if chunks:
        r = bt.results()                                   # flushes the last decode
        ids_l.append(r["ids"][:len(chunks[-1])]); lens_l.append(r["lens"][:len(chunks[-1])])
    if barrier:
        barrier()
    elapsed = time.perf_counter() - t0
    ids_loc = np.concatenate(ids_l) if ids_l else np.zeros((0, 1), np.int32)
    lens_loc = np.concatenate(lens_l) if lens_l else np.zeros(0, np.int32)
    mt = ids_loc.shape[1]
    if world > 1:
        import torch
        t = torch.tensor([mt], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mtg = int(t.item())
        if mtg > mt:
            ids_loc = np.pad(ids_loc, ((0, 0), (0, mtg - mt)))
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ids, lens = shard.gather_token_matrix(ids_loc, lens_loc, idx, n_clips, world, dist, device=f"cuda:{local_rank}")
    else:
        ids, lens = shard.gather_token_matrix(ids_loc, lens_loc, idx, n_clips, 1)
    bt.close()
    model.close()
    return ids, lens, elapsed


def run_mixed(n_clips, lo_s, hi_s, decoder, cfg_name="tdt-ctc-110m", layers=None, dist=None, rank=0, world=1, local_rank=0, barrier=None):
    """Mixed-length corpus (round 4): clip i has a seeded random length in [lo_s, hi_s] seconds; shard.shard_by_audio deals the clips by audio,
    every rank runs ITS clips through pk_transcribe_pcm (sorted, packed into ragged batches, two-stream pipeline; uploads and results inside
    the clock), one all_gather of the token matrix at the end.  The corpus is a pure function of the clip index: any world size transcribes
    the same audio and the digests are comparable."""
    import dataclasses
    import numpy as np
    import pkload
    pk = pkload.load()
    from parakeet_cpp_amd import capi, shard, synth
    cfg = pk.PRESETS[cfg_name]()
    if layers:
        cfg = dataclasses.replace(cfg, num_layers=layers, name=f"{cfg.name}-{layers}L")
    wpath = f"/tmp/pk_sharded_{cfg.name}_{cfg.num_layers}_seed42.safetensors"
    if local_rank == 0 and not os.path.exists(wpath):
        synth.save_weights(wpath + f".{os.getpid()}", synth.synth_weights(cfg, seed=42))
        os.replace(wpath + f".{os.getpid()}", wpath)
    if barrier:
        barrier()
    model = capi.Model(wpath, cfg, device=local_rank)
    rng = np.random.default_rng(2026)
    lengths = [int(x) for x in rng.uniform(lo_s, hi_s, n_clips) * 16000]
    idx = shard.shard_by_audio(lengths, rank, world)
    POOL = 16                                              # distinct base clips at the longest length; clip i = a window of base clip i % POOL
    base = synth.synth_pcm(POOL, int(hi_s * 16000) + 1, seed=4321)
    clips = [base[i % POOL][(i * 37) % 97:(i * 37) % 97 + lengths[i]] for i in idx]
    packed = capi.pack_clips(clips) if clips else None      # the C ABI's input form (pcm + offsets), built before the clock starts
    if clips:
        model.transcribe_pcm(clips[:2], decoder=decoder)   # the pipeline's buffers exist before the clock starts
    if barrier:
        barrier()
    t0 = time.perf_counter()
    res = model.transcribe_pcm(packed, decoder=decoder) if clips else []
    if barrier:
        barrier()
    elapsed = time.perf_counter() - t0
    mt = max([len(r["token_ids"]) for r in res] + [1])
    if world > 1:
        import torch
        t = torch.tensor([mt], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mt = int(t.item())
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ids_loc = np.zeros((len(res), mt), np.int32)
    for k, r in enumerate(res):
        ids_loc[k, :len(r["token_ids"])] = r["token_ids"]
    lens_loc = np.asarray([len(r["token_ids"]) for r in res], np.int32)
    ids, lens = shard.gather_token_matrix(ids_loc, lens_loc, idx, n_clips, world, dist if world > 1 else None,
                                          device=f"cuda:{local_rank}" if world > 1 else None)
    model.close()
    return ids, lens, elapsed, sum(lengths) / 16000.0


def digest(ids, lens):
    h = hashlib.sha256()
    for i in range(len(lens)):
        h.update(ids[i, :lens[i]].tobytes())
        h.update(b"|")
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--clip-seconds", type=float, default=10.0)
    ap.add_argument("--decoder", default="tdt", choices=["tdt", "ctc"])
    ap.add_argument("--config", default="tdt-ctc-110m")
    ap.add_argument("--layers", type=int, default=0, help="cut the encoder to this many layers (tests)")
    ap.add_argument("--broadcast-weights", action="store_true", help="rank 0 reads the weights once and broadcasts the image (RCCL)")
    ap.add_argument("--mixed", type=float, nargs=2, metavar=("LO_S", "HI_S"), help="mixed-length corpus: clip lengths uniform in [LO_S, HI_S] seconds, "
                    "sharded by audio, packed into ragged batches on every rank")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mixed:
        ids, lens, elapsed, audio_s = run_mixed(args.clips, args.mixed[0], args.mixed[1], args.decoder, args.config, args.layers or None,
                                                dist if world > 1 else None, rank, world, local_rank, barrier)
        if rank == 0:
            print(json.dumps({"workload": f"{args.config}: {args.clips} clips of {args.mixed[0]:g}-{args.mixed[1]:g} s (mixed lengths, packed), {args.decoder.upper()} greedy",
                              "n_gpus": world, "audio_s": round(audio_s, 1), "wall_s": round(elapsed, 4), "rtfx": round(audio_s / elapsed, 1),
                              "tokens": int(lens.sum()), "digest": digest(ids, lens),
                              "note": "sharded by audio (shard.shard_by_audio), every rank's clips through pk_transcribe_pcm: sort, pack, PCIe, pipeline, results inside the clock"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    n = int(args.clip_seconds * 16000)
    ids, lens, elapsed = run(args.clips, args.batch, n, args.decoder, args.config, args.layers or None, dist if world > 1 else None, rank, world,
                             local_rank, barrier, args.broadcast_weights)
    if rank == 0:
        print(json.dumps({"workload": f"{args.config}: {args.clips} x {args.clip_seconds:g} s clips, batches of {args.batch}, {args.decoder.upper()} greedy",
                          "n_gpus": world, "clips_per_gpu": (args.clips + world - 1) // world, "wall_s": round(elapsed, 4),
                          "rtfx": round(args.clips * args.clip_seconds / elapsed, 1), "tokens": int(lens.sum()), "digest": digest(ids, lens),
                          "note": "wall includes the H2D upload of every batch (copy stream, under the previous encoder) and the D2H of every result"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
