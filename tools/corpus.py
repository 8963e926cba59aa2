#!/usr/bin/env python3
"""Run BASELINE.json's sharded configs for real: every clip through shard.assign_pieces + shard.run_shard,
one process per rank, distinct synthetic frames per clip, plan switches included.

  configs[3]  8 x 600-frame 3840x2160 clips -> portrait 1080x1920         (--config 3)
  configs[4]  100 clips, 34 x 720p / 33 x 1080p / 33 x 4K, 300 frames each  (--config 4)
  reduced     2 + 2 + 2 clips of 30 frames at the same sizes                (--config reduced: a rehearsal)

The reference's loop over clips is sequential (scripts/processVideos.js:167-169); here rank r takes the pieces
assign_pieces() gives it.  Ranks do not exchange frames: torch.distributed (gloo, CPU) carries the closing
barrier and the gather of per-rank results only, so several ranks may share one GPU for a rehearsal
(--ranks-per-gpu; device = local rank mod visible GPUs).  Started without a launcher, `--gpus N` starts the N
ranks itself, before this process touches the GPU.

Prints one JSON object: per rank {pieces, frames, scale_s = device time of the scaler launches, wall_s = the
rank's whole loop, frame generation included}, the max over ranks of both, frames/s = all frames / max scale_s,
and a checksum over all clips that does not depend on the number of ranks or on the GOP cut.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VF = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
OUT = (1080, 1920)
SIZES = {"720p": (1280, 720), "1080p": (1920, 1080), "4k": (3840, 2160)}


def corpus(config):
    from tiktoken_amd.shard import Clip
    if config == "3":
        return [Clip("4k60_%d" % i, 3840, 2160, 600) for i in range(8)]
    if config == "4":
        return [Clip("clip%03d" % i, *(SIZES["720p"] if i % 3 == 0 else SIZES["1080p"] if i % 3 == 1 else SIZES["4k"]), 300)
                for i in range(100)]
    if config == "reduced":
        return [Clip("r%d" % i, *list(SIZES.values())[i % 3], 30) for i in range(6)]
    raise SystemExit("corpus.py: --config 3 | 4 | reduced")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="reduced")
    ap.add_argument("--gpus", type=int, default=1, help="ranks")
    ap.add_argument("--gop", type=int, default=30)
    ap.add_argument("--batch", type=int, default=150, help="frames per launch (and per generated buffer)")
    ap.add_argument("--ranks-per-gpu", type=int, default=1, help="> 1: rehearse more ranks than GPUs")
    ap.add_argument("--pool", type=int, default=0, help="> 0: frames come from a pool of this many pre-generated frames per source size "
                                                        "(made before the clock starts; clip i reads it from an offset of its own), so the wall "
                                                        "time is the driver's -- plans, launches, checksums -- not the synthetic decoder's")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch
        have = torch.cuda.device_count()                  # (does not initialise the GPU)
        if have * args.ranks_per_gpu < args.gpus:
            raise SystemExit("corpus.py: --gpus %d but %d GPU(s) x %d rank(s) each" % (args.gpus, have, args.ranks_per_gpu))
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist
    import tiktoken_amd as tts
    from tiktoken_amd.shard import assign_pieces, piece_imbalance, run_shard

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("corpus.py: WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    torch.cuda.set_device(local % torch.cuda.device_count())
    from tiktoken_amd.shard import gpu_pci_address, pin_to_gpu_numa
    pinned = pin_to_gpu_numa(gpu_pci_address(torch.cuda.current_device()))      # host threads next to the GPU's root complex
    if world > 1:
        dist.init_process_group("gloo")

    clips = corpus(args.config)
    shards = assign_pieces(clips, world, OUT, gop=args.gop)

    gen = torch.Generator(device="cuda")

    def make_frames(c, first, n):
        # a frame's content depends on (clip, frame number) alone, so the corpus checksum is the same for any
        # number of ranks, any GOP cut and any batch size
        nbytes = tts.frame_bytes(c.width, c.height)
        out = torch.empty((n, nbytes), dtype=torch.uint8, device="cuda")
        for k in range(n):
            gen.manual_seed(zlib.crc32(("%s/%d" % (c.name, first + k)).encode()))
            out[k] = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda", generator=gen)
        return out

    pools = {}
    if args.pool > 0:
        for c in clips:
            if (c.width, c.height) not in pools:
                nbytes = tts.frame_bytes(c.width, c.height)
                gen.manual_seed(c.width * 65536 + c.height)
                pools[(c.width, c.height)] = torch.randint(0, 256, (args.pool + args.batch, nbytes), dtype=torch.uint8, device="cuda", generator=gen)
        index = {c.name: i for i, c in enumerate(clips)}
        plain_make = make_frames

        def make_frames(c, first, n):                    # noqa: F811  (a view: nothing is generated or copied inside the clock)
            if n > args.batch:
                return plain_make(c, first, n)
            start = (first + 37 * index[c.name]) % args.pool
            return pools[(c.width, c.height)][start:start + n]

    # warm: build the plans, settle the clocks (untimed)
    run_shard(clips, [p for p in shards[rank]][:3], VF, lambda c, f, n: make_frames(c, f, min(n, 8)), batch=8)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timing = {}
    t0 = time.perf_counter()
    sums = run_shard(clips, shards[rank], VF, make_frames, batch=args.batch, timing=timing, checksum=not args.pool)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if args.pool:
        # pool mode times the driver alone (plans, launches); the checksums -- torch reductions that take ten times the
        # scaler's time (tools/walltime.py) -- come from a second, untimed pass over the same pieces
        sums = run_shard(clips, shards[rank], VF, make_frames, batch=args.batch)
    mine = {"rank": rank, "pieces": len(shards[rank]), "frames": timing.get("frames", 0),
            "scale_s": round(timing.get("scale_s", 0.0), 6), "wall_s": round(wall, 4), "numa_cpus": len(pinned), "sums": sums}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        total = {}
        for r in ranks:
            for k, v in r.pop("sums").items():
                total[k] = total.get(k, 0) + v
        frames = sum(r["frames"] for r in ranks)
        max_scale, max_wall = max(r["scale_s"] for r in ranks), max(r["wall_s"] for r in ranks)
        print(json.dumps({
            "config": args.config, "clips": len(clips), "ranks": world, "gop": args.gop, "batch": args.batch, "pool": args.pool,
            "frames": frames, "expected_frames": sum(c.frames for c in clips),
            "imbalance": round(piece_imbalance(clips, shards, OUT), 4),
            "max_scale_s": max_scale, "max_wall_s": max_wall,
            "frames_per_s": round(frames / max_scale, 1) if max_scale else None,
            "frames_per_s_wall": round(frames / max_wall, 1) if max_wall else None,
            "checksum": "%016x" % (zlib.crc32(json.dumps(sorted(total.items())).encode()) | (sum(total.values()) & 0xFFFFFFFF) << 32),
            "per_rank": ranks}), flush=True)


if __name__ == "__main__":
    main()
