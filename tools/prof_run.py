#!/usr/bin/env python3
"""Tiny workload for rocprofv3 passes: a few batches of one sweep case (env knobs apply)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiktoken_amd as tts
from sweep import CASES, plan_for

case = sys.argv[1] if len(sys.argv) > 1 else "portrait"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
vf, iw, ih, _ = CASES[case]
plan = plan_for(case)
src = torch.randint(0, 256, (frames, plan.in_bytes), dtype=torch.uint8, device="cuda")
dst = torch.empty((frames, plan.out_bytes), dtype=torch.uint8, device="cuda")
for _ in range(iters):
    plan.run_batch(src, dst)
torch.cuda.synchronize()
