#!/usr/bin/env python3
"""bench.py -- frames/s of the 1080p -> 1080x1920 portrait scale+crop hot path on MI355X.

One step = one pass of libttscale over one HBM-resident batch of synthetic yuv420p frames
(BASELINE.json configs[1] geometry, batched: 1,350 frames = 45 s of 1080p30, 4.2 GB of source and
4.2 GB of output per GPU, one kernel launch; `--frames 300` is the 10 s clip of configs[0]: the
launch's ramp and tail weigh more there, 906 k frames/s against 990 k).
Multi-GPU: one process per GPU, every rank owns its own clip (clip-sharded, no data-path
collective; torch.distributed is used for the barrier and the max-over-ranks only).

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for the byte accounting.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VF = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
IN_W, IN_H = 1920, 1080
# minimum source window read once (608x1080 luma window + chroma) + output written once
ALGO_BYTES_PER_FRAME = 4_095_360
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(geom, budget_s=12.0):
    """The CPU oracle (own restatement of the swscale C path, NOT ffmpeg), one thread, timed on a
    bounded sample of the same workload.  Like ffmpeg it scales the whole 3413x1920 frame, then crops."""
    from oracle import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    src = synth.frames("random", IN_W, IN_H, 2)
    oracle.run_chain(geom, src[:1])            # warm (page in, build tables once)
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.run_chain(geom, src[n % 2: n % 2 + 1])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and n >= 4 or n >= 2000:
            break
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames 1920x1080 -> 3413x1920 bicubic -> crop 1080x1920, oracle/swscale_oracle.c -O2, "
                      "1 thread, %.1f s" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=1350, help="frames per GPU per step (1,350 x 3.11 MB = 4.2 GB: the most that "
                                                             "fits one set of launch-wide buffer descriptors, DESIGN.md 3)")
    ap.add_argument("--settle", type=int, default=20,
                    help="untimed launches BEFORE the warm-up steps: back-to-back launch times drift 0.33 -> 0.42 -> 0.35 ms over "
                         "the first ~30 launches of a burst while the clock governor settles (DESIGN.md 4.5); the metric is sustained rate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import tiktoken_amd as tts

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("TTS_BENCH_FORCE_DIST"):      # (the env knob rehearses the RCCL path on one GPU)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    plan = tts.Plan.from_vf(VF, IN_W, IN_H)
    assert plan.kernel_kind(0) == 2 and plan.kernel_kind(1) == 2, "hot path must use the LDS-staged strip kernel"
    n = args.frames
    gen = torch.Generator(device="cuda").manual_seed(0x5EED + rank)
    src = torch.randint(0, 256, (n, plan.in_bytes), dtype=torch.uint8, device="cuda", generator=gen)
    dst = torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda")

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.settle + args.warmup):
        plan.run_batch(src, dst)
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        plan.run_batch(src, dst)
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0              # this rank's K steps; the MAX over ranks is taken below
    sync()                                       # closing bracket: barrier + synchronize
    dev_ms = ev0.elapsed_time(ev1)               # HIP events on the launch stream (torch's current stream)

    if dist is not None:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    if rank == 0:
        ms_step = wall * 1e3 / args.steps
        fps = world * n * args.steps / wall
        # A step is ONE launch of the strip kernel (luma and chroma planes ride in one grid), so its
        # average launch duration is the event time per step and a launch moves a whole step's bytes.
        launches = 1                               # (tts_run_batch covers a batch of any length with one launch)
        launch_ms = dev_ms / args.steps / launches
        algo_bytes_launch = ALGO_BYTES_PER_FRAME * n / launches
        achieved = algo_bytes_launch / (launch_ms * 1e-3) / 1e9
        traffic = None                             # PMC-measured HBM bytes per frame (profiles/traffic.json) x frames per launch
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = int(json.load(open(tpath))["hbm_bytes_per_frame"] * n / launches)
            except Exception:
                traffic = None
        out = {
            "metric": "frames/sec 1080p->1080x1920 portrait scale+crop",
            "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1920x1080 yuv420p -> scale 3413x1920 bicubic -> crop 1080x1920, "
                                   "%d HBM-resident frames per GPU per step" % n,
                       "vf": VF, "frames_per_gpu": n, "settle_launches": args.settle, "parallelism": "clip-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "strip_lds_kernel<4, 4, 4, 1, 1>", "launch_ms": round(launch_ms, 4),
                         "algo_bytes_per_launch": int(algo_bytes_launch)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(plan.geom)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
