#!/usr/bin/env python3
"""Where a corpus run's wall time goes: the pieces of shard.run_shard's loop timed apart (150 frames of 1080p a batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiktoken_amd as tts
from tiktoken_amd.shard import _checksum

VF = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
plan = tts.Plan.from_vf(VF, 1920, 1080)
n = 150
src = torch.randint(0, 256, (n, plan.in_bytes), dtype=torch.uint8, device="cuda")
dst = torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda")


def timed(label, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s host %.3f ms a call, with the GPU drained %.3f ms" % (label, (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3))


timed("run_batch into a given tensor", lambda: plan.run_batch(src, dst))
timed("run_batch allocating its output", lambda: plan.run_batch(src))
timed("checksum (dword lanes)", lambda: _checksum(dst))
timed("checksum (torch uint8 -> int64 sum)", lambda: dst.sum(dtype=torch.int64))
timed("two events", lambda: (torch.cuda.Event(enable_timing=True).record(), torch.cuda.Event(enable_timing=True).record()))
timed("torch.empty of the output", lambda: torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda"))
