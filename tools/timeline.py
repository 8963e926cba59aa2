#!/usr/bin/env python3
"""Workgroup timeline of one strip_lds_kernel launch (experiments build: make -C tiktoken_amd/csrc exp;
run with TTS_LIB=tiktoken_amd/libttscale_exp.so TTS_EXP_TIMELINE=1).  Prints, in 10 us bins from the first
workgroup's start, how many workgroups are running, and the distribution of workgroup durations by start time.
Usage: python tools/timeline.py [case] [frames]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiktoken_amd as tts
from sweep import CASES, plan_for

case = sys.argv[1] if len(sys.argv) > 1 else "portrait"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
vf, iw, ih, algo = CASES[case]
plan = plan_for(case)
src = torch.randint(0, 256, (frames, plan.in_bytes), dtype=torch.uint8, device="cuda")
dst = torch.empty((frames, plan.out_bytes), dtype=torch.uint8, device="cuda")
for _ in range(30):
    plan.run_batch(src, dst)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); plan.run_batch(src, dst); e1.record(); torch.cuda.synchronize()
L = ctypes.CDLL(tts.LIB_PATH)
buf = np.zeros((1 << 20, 2), np.uint64)
n = L.tts_exp_timeline(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(1 << 20))
assert n > 0, "no timeline: experiments build + TTS_EXP_TIMELINE=1 needed"
t = buf[:n].astype(np.int64)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # us (100 MHz clock)
print("%s, %d frames: %d workgroups, launch %.1f us by events, first start -> last end %.1f us" % (case, frames, n, e0.elapsed_time(e1) * 1e3, end.max()))
dur = end - start
print("workgroup duration us: min %.1f  median %.1f  max %.1f" % (dur.min(), np.median(dur), dur.max()))
edges = np.arange(0, end.max() + 10, 10)
running = [(np.count_nonzero((start < b + 10) & (end > b))) for b in edges]
print("running workgroups per 10 us bin:")
print(" ".join("%d" % r for r in running))
order = np.argsort(start)
q = np.array_split(order, 8)
print("by start-time octile: mean start us / mean duration us")
print("  ".join("%.0f/%.0f" % (start[i].mean(), dur[i].mean()) for i in q))
