#!/usr/bin/env python3
"""1080p -> portrait with padded output rows: python tools/strided.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tiktoken_amd as tts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
plan = tts.Plan.from_vf("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1920, 1080)
src = torch.randint(0, 256, (n, plan.in_bytes), dtype=torch.uint8, device="cuda")
for ds in ((1080, 540), (1088, 544), (1088, 576), (1152, 576)):
    dp = plan.strided_bytes("out", *ds)
    dst = torch.empty((n, dp), dtype=torch.uint8, device="cuda")
    for _ in range(30): plan.run_batch_strided(src, (1920, 960), dst, ds, n)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.run_batch_strided(src, (1920, 960), dst, ds, n)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    ts.sort()
    print("dst strides %s: %.4f ms per %d frames, %.0f frames/s, %.1f %% of 8 TB/s (algorithmic bytes)" % (
        ds, ts[3], n, n / ts[3] * 1e3, 4095360 * n / ts[3] / 1e6 / 80), flush=True)
    del dst
