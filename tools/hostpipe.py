#!/usr/bin/env python3
"""PCIe-inclusive rate of tts_run_host (pageable numpy buffers in and out): python tools/hostpipe.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tiktoken_amd as tts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
plan = tts.Plan.from_vf("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1920, 1080)
host = np.random.default_rng(5).integers(0, 256, (n, plan.in_bytes), dtype=np.uint8)
plan.run_host(host[:4])
def show(tag, src, out):
    for rep in range(3):
        t0 = time.perf_counter(); plan.run_host(src, out); dt = time.perf_counter() - t0
        print("%-28s CHUNK_MB=%s: %d frames %.1f ms  %.0f frames/s  %.1f GB/s each way" % (
            tag, os.environ.get("TTS_HOST_CHUNK_MB", "128"), n, dt * 1e3, n / dt, n * plan.in_bytes / dt / 1e9), flush=True)
show("pageable, fresh output", host, None)
out = np.zeros((n, plan.out_bytes), np.uint8)
show("pageable, touched output", host, out)
import torch
pin_in = torch.empty((n, plan.in_bytes), dtype=torch.uint8).pin_memory(); pin_in.copy_(torch.from_numpy(host))
pin_out = torch.empty((n, plan.out_bytes), dtype=torch.uint8).pin_memory()
show("pinned by the caller", pin_in.numpy(), pin_out.numpy())
assert np.array_equal(pin_out.numpy(), out)
