#!/usr/bin/env python3
"""BASELINE.json configs[2..4] and the PCIe-inclusive rate on ONE GPU.

  config 3  10,000 frames 1080x1920 -> 320x568 `area`, all 31 GB HBM-resident
  config 4  one rank's share of "4K60 clips -> portrait": 600 frames 3840x2160
  config 5  the mixed corpus (34x720p, 33x1080p, 33x4K clips of 300 frames): every clip is run once
            on this GPU; the 2/4/8-GPU makespans follow from tiktoken_amd.shard.assign over the
            measured per-clip times (clips are independent, there is no collective to model)
  pcie      30 frames of the bench job through tts_run_host (pageable host buffers in and out)

Prints one JSON object; `python tools/configs.py > gpurun_out/configs.json`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tiktoken_amd as tts
from tiktoken_amd.shard import Clip, assign

PORTRAIT = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
THUMB = "scale=320:568:flags=area"


def rand_frames(n, nbytes, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty((n, nbytes), dtype=torch.uint8, device="cuda")
    step = max(1, (1 << 30) // nbytes)
    for i in range(0, n, step):
        m = min(step, n - i)
        out[i:i + m] = torch.randint(0, 256, (m, nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    return out


def timed(plan, src, dst, reps, warm=2):
    for _ in range(warm):
        plan.run_batch(src, dst)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run_batch(src, dst)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    res = {}
    # ---- config 3
    plan = tts.Plan.from_vf(THUMB, 1080, 1920)
    n = 10000
    src = rand_frames(n, plan.in_bytes, 3)
    dst = torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda")
    ms = timed(plan, src, dst, 5)
    res["config3_thumbs_10k"] = {"frames": n, "ms": round(ms, 3), "frames_per_s": round(n / ms * 1e3),
                                 "algorithmic_GBps": round(3383040 * n / ms / 1e6, 1), "frac_of_8TBps": round(3383040 * n / ms / 1e6 / 8000, 4),
                                 "hbm_resident_GB": round((src.numel() + dst.numel()) / 1e9, 1)}
    del src, dst
    torch.cuda.empty_cache()
    # ---- config 4 (per GPU: one 600-frame 4K clip)
    plan = tts.Plan.from_vf(PORTRAIT, 3840, 2160)
    n = 600
    src = rand_frames(n, plan.in_bytes, 4)
    dst = torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda")
    ms = timed(plan, src, dst, 10)
    res["config4_4k_clip_600"] = {"frames": n, "ms": round(ms, 3), "frames_per_s": round(n / ms * 1e3),
                                  "algorithmic_GBps": round(7050240 * n / ms / 1e6, 1), "frac_of_8TBps": round(7050240 * n / ms / 1e6 / 8000, 4)}
    del src, dst
    torch.cuda.empty_cache()
    # ---- config 5
    sizes = [(1280, 720, 34), (1920, 1080, 33), (3840, 2160, 33)]
    clips, per_clip_ms = [], {}
    for w, h, cnt in sizes:
        plan = tts.Plan.from_vf(PORTRAIT, w, h)
        src = rand_frames(300, plan.in_bytes, w)
        dst = torch.empty((300, plan.out_bytes), dtype=torch.uint8, device="cuda")
        per_clip_ms[(w, h)] = timed(plan, src, dst, 20, warm=5)
        clips += [Clip("%dx%d_%02d" % (w, h, i), w, h, 300) for i in range(cnt)]
        del src, dst
    total_ms = sum(per_clip_ms[(c.width, c.height)] for c in clips)
    c5 = {"clips": len(clips), "frames": 300 * len(clips),
          "ms_per_clip": {"%dx%d" % k: round(v, 4) for k, v in per_clip_ms.items()},
          "n1_ms": round(total_ms, 2), "n1_frames_per_s": round(300 * len(clips) / total_ms * 1e3)}
    for world in (2, 4, 8):
        shards = assign(clips, world, (1080, 1920))
        loads = [sum(per_clip_ms[(clips[i].width, clips[i].height)] for i in s) for s in shards]
        c5["n%d_makespan_ms_from_measured_clip_times" % world] = round(max(loads), 2)
        c5["n%d_efficiency" % world] = round(total_ms / world / max(loads), 4)
    res["config5_mixed_corpus"] = c5
    # ---- PCIe-inclusive
    plan = tts.Plan.from_vf(PORTRAIT, 1920, 1080)
    n = 30
    host = np.random.default_rng(5).integers(0, 256, (n, plan.in_bytes), dtype=np.uint8)
    plan.run_host(host[:2])
    t0 = time.perf_counter()
    plan.run_host(host)
    dt = time.perf_counter() - t0
    res["pcie_inclusive_run_host"] = {"frames": n, "frames_per_s": round(n / dt), "note": "pageable host buffers, hipMalloc + H2D + kernel + D2H + hipFree per call"}
    pin_in = torch.empty((n, plan.in_bytes), dtype=torch.uint8).pin_memory()
    pin_out = torch.empty((n, plan.out_bytes), dtype=torch.uint8).pin_memory()
    pin_in.copy_(torch.from_numpy(host))
    d_in = torch.empty((n, plan.in_bytes), dtype=torch.uint8, device="cuda")
    d_out = torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d_in.copy_(pin_in, non_blocking=True)
        plan.run_batch(d_in, d_out)
        pin_out.copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res["pcie_inclusive_pinned"] = {"frames": n, "frames_per_s": round(n / dt),
                                    "GBps_each_way": round(n * plan.in_bytes / dt / 1e9, 1), "note": "pinned buffers, one stream, no overlap"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
