#!/usr/bin/env python3
"""Launch time against frames per launch for one sweep case: is the gap between small and large launches a
fixed cost per launch (intercept of a straight line) or a staircase (workgroup rounds)?
Usage: python tools/scaling.py [case] [comma list of frame counts]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiktoken_amd as tts
from sweep import CASES

case = sys.argv[1] if len(sys.argv) > 1 else "portrait"
counts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "32,64,100,150,200,256,300,350,400,450,512,600,800,1000,1350").split(",")]
vf, iw, ih, algo = CASES[case]
plan = tts.Plan.from_vf(vf, iw, ih)
nmax = max(counts)
src = torch.randint(0, 256, (nmax, plan.in_bytes), dtype=torch.uint8, device="cuda")
dst = torch.empty((nmax, plan.out_bytes), dtype=torch.uint8, device="cuda")
for _ in range(30):
    plan.run_batch(src[:300], dst[:300])
res = {n: [] for n in counts}
for rnd in range(6):
    for n in counts:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.run_batch(src[:n], dst[:n])
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[n].append(e0.elapsed_time(e1) / 10)
print("case %s: frames, ms per launch (median), us per frame, %% of 8 TB/s" % case)
for n in counts:
    t = sorted(res[n])[len(res[n]) // 2]
    print("  %5d  %8.4f  %7.4f  %5.1f%%" % (n, t, t * 1e3 / n, algo * n / (t * 1e-3) / 8e12 * 100))
