#!/usr/bin/env python3
"""bench.py -- frames/s of the 1080p -> 1080x1920 portrait scale+crop hot path on MI355X.

One step = one pass of libttscale over one HBM-resident batch of synthetic yuv420p frames
(BASELINE.json configs[1] geometry, batched: 1,350 frames = 45 s of 1080p30, 4.2 GB of source and
4.2 GB of output per GPU, one kernel launch; `--frames 300` is the 10 s clip of configs[0]: the
launch's ramp and tail weigh more there).
Multi-GPU: one process per GPU, every rank owns its own clip (clip-sharded, no data-path
collective; torch.distributed is used for the barrier and the max-over-ranks only).  Started
without a launcher (`python bench.py --gpus 4`), it starts the ranks itself -- BEFORE anything in this
process touches the GPU -- through `python -m torch.distributed.run`, and exits with their status.

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for the byte accounting.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VF = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
IN_W, IN_H = 1920, 1080
# minimum source window read once (608x1080 luma window + chroma) + output written once
ALGO_BYTES_PER_FRAME = 4_095_360
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
def kernel_sources():
    """Every file the shipped library is built from (csrc/Makefile: HIPSRCS, CXXSRCS, HDRS): a change in any of them --
    chunking in ttscale.hip, a tap table in filter.cpp, an RGB kernel -- may alter launches and traffic."""
    d = os.path.join(ROOT, "tiktoken_amd", "csrc")
    names = sorted(n for n in os.listdir(d) if n.endswith((".hip", ".h", ".cpp")) and n != "ttscale_cli.cpp")
    return [os.path.join(d, n) for n in names] + [os.path.join(ROOT, "include", "ttscale.h")]


def kernel_sources_hash():
    """What profiles/traffic.json is stamped with: the PMC figure is only quoted for the code it was measured on."""
    h = hashlib.sha256()
    for path in kernel_sources():
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def visible_gpus():
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the parent of the ranks must not initialise
    the GPU; torch.cuda.device_count() falls back to hipGetDeviceCount where amdsmi is missing): KFD's topology in
    sysfs lists one node per device, GPUs being those with SIMDs; a *_VISIBLE_DEVICES list narrows it."""
    n = 0
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(top):
            try:
                with open(os.path.join(top, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                n += int(props.get("simd_count", "0")) > 0
            except OSError:
                pass
    except OSError:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _port_worker(args):
    """One process of the all-cores leg: frames through the oracle until the deadline."""
    geom_dict, seed, deadline = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from oracle import oracle
    src = synth.frames("random", IN_W, IN_H, 1, seed=seed)
    n = 0
    while time.perf_counter() < deadline or n == 0:
        oracle.run_chain(geom_dict, src)
        n += 1
    return n


def cpu_baseline(geom, budget_s=9.0):
    """CPU side of the comparison, on this box's host cores, on a bounded sample of the same workload.
    A real `ffmpeg` on PATH is used if there is one (kind "ffmpeg": the reference's own engine, -threads 1
    and all cores); else the CPU oracle (own restatement of the swscale C path, NOT ffmpeg; kind "port"),
    one thread and one process per core.  Like ffmpeg the oracle scales the whole 3413x1920 frame, then crops."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))      # the box's CPU share, not the machine's
    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg:
        n = 60
        path = "/dev/shm/tts_bench_%d.yuv" % os.getpid()
        synth.frames("random", IN_W, IN_H, 4).tofile(path)
        res = {}
        try:
            for label, threads in (("one", 1), ("all", cores)):
                cmd = [ffmpeg, "-v", "error", "-threads", str(threads), "-filter_threads", str(threads), "-stream_loop", str(n // 4 - 1),
                       "-f", "rawvideo", "-pix_fmt", "yuv420p", "-s", "%dx%d" % (IN_W, IN_H), "-i", path,
                       "-vf", VF, "-f", "rawvideo", "-y", os.devnull]
                t0 = time.perf_counter()
                subprocess.run(cmd, check=True, timeout=120)
                res[label] = n / (time.perf_counter() - t0)
        finally:
            os.unlink(path)
        return {"value": round(res["one"], 3), "unit": "frames/s", "cores": 1, "kind": "ffmpeg",
                "all_cores": {"value": round(res["all"], 3), "cores": cores},
                "sample": "%d frames 1920x1080 rawvideo from /dev/shm -> -vf %s -> /dev/null, %s" % (n, VF, ffmpeg)}
    from oracle import oracle
    src = synth.frames("random", IN_W, IN_H, 2)
    oracle.run_chain(geom, src[:1])            # warm (page in, build tables once)
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.run_chain(geom, src[n % 2: n % 2 + 1])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and n >= 4 or n >= 2000:
            break
    out = {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames 1920x1080 -> 3413x1920 bicubic -> crop 1080x1920, oracle/swscale_oracle.c -O2, "
                     "1 thread, %.1f s; all_cores: one process per core for the same time" % (n, dt)}
    try:
        import multiprocessing as mp
        gd = geom.as_dict()
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(cores) as pool:
            counts = pool.map(_port_worker, [(gd, 100 + i, time.perf_counter() + budget_s) for i in range(cores)])
        out["all_cores"] = {"value": round(sum(counts) / (time.perf_counter() - t0), 3), "cores": cores}
    except Exception as e:                      # the one-thread figure stands on its own
        out["all_cores"] = {"error": str(e)}
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) and relay their output.
    Nothing here may initialise the GPU: the devices are counted from sysfs (visible_gpus)."""
    import socket
    have = visible_gpus()
    if have is not None and have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s); refusing to report n_gpus=%d for fewer devices\n"
                         % (args.gpus, have, args.gpus))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))              # (this process never touches the GPU)
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import numpy as np
    import torch
    import tiktoken_amd as tts

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    # Outside the timed region: two frames of the batch the timed launches wrote, against the oracle.
    from oracle import oracle
    checked = sorted({0, n - 1})
    for i in checked:
        want = oracle.run_chain(plan.geom, src[i:i + 1].cpu().numpy())
        if not np.array_equal(dst[i:i + 1].cpu().numpy(), want):
            raise SystemExit("bench.py: rank %d: frame %d of the timed batch differs from the oracle" % (rank, i))

    if dist is not None:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    if rank == 0:
        ms_step = wall * 1e3 / args.steps
        fps = world * n * args.steps / wall
        # A step is ONE launch of the strip kernel (luma and chroma planes ride in one grid), so its
        # average launch duration is the event time per step and a launch moves a whole step's bytes.
        launches = 1                               # (tts_run_batch covers a batch of any length with one launch: both classes report the same kernel)
        assert plan.kernel_name(0) == plan.kernel_name(1)
        launch_ms = dev_ms / args.steps / launches
        algo_bytes_launch = ALGO_BYTES_PER_FRAME * n / launches
        achieved = algo_bytes_launch / (launch_ms * 1e-3) / 1e9
        # PMC-measured HBM bytes per frame (profiles/traffic.json, tools/traffic.py) x frames per launch: quoted
        # only while the kernel sources are the ones it was measured on
        traffic, traffic_note = None, "no profiles/traffic.json"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_sources_sha256") == kernel_sources_hash():
                    traffic = int(tj["hbm_bytes_per_frame"] * n / launches)
                    traffic_note = "PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE of %s, scaled per frame" % tj.get("source", "profiles/")
                else:
                    traffic_note = "profiles/traffic.json was measured on other kernel sources (hash differs): not quoted"
            except Exception as e:
                traffic_note = "profiles/traffic.json unreadable: %s" % e
        out = {
            "metric": "frames/sec 1080p->1080x1920 portrait scale+crop",
            "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1920x1080 yuv420p -> scale 3413x1920 bicubic -> crop 1080x1920, "
                                   "%d HBM-resident frames per GPU per step" % n,
                       "vf": VF, "frames_per_gpu": n, "settle_launches": args.settle, "parallelism": "clip-sharded x%d, no collective" % world,
                       "checked_frames_vs_oracle": checked},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": plan.kernel_name(0), "launches_per_step": launches, "launch_ms": round(launch_ms, 4),
                         "algo_bytes_per_launch": int(algo_bytes_launch)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(plan.geom)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
