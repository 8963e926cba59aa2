#!/usr/bin/env python3
"""rocprofv3 workload: the reference's thumbnail chain with the mjpeg range pass
(src/services/video.js:329-334), one call per thumbnail and one batch -- to show what a call launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiktoken_amd as tts

VF = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p,format=yuvj420p"
for iw, ih in ((720, 1280), (1920, 1080)):
    plan = tts.Plan.from_vf(VF, iw, ih)
    one = torch.randint(0, 256, (1, plan.in_bytes), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        plan.run_batch(one)
    many = torch.randint(0, 256, (300, plan.in_bytes), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        plan.run_batch(many)
torch.cuda.synchronize()
