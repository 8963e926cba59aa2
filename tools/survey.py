#!/usr/bin/env python3
"""Which kernel does every common geometry take, and how fast is it?  python tools/survey.py [frames]
Algorithmic bytes = whole source read once + output written once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tiktoken_amd as tts
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
SIZES = [(640, 360), (1280, 720), (1920, 1080), (3840, 2160), (720, 1280), (1080, 1920)]
print("%-11s %-11s %-9s %-30s %8s %6s" % ("src", "dst", "alg", "kernel (luma)", "ms", "%peak"))
for iw, ih in SIZES:
    for ow, oh in SIZES + [(320, 180), (180, 320)]:
        if (ow, oh) == (iw, ih) or (ow > oh) != (iw > ih):
            continue
        for alg in ("bicubic", "area", "bilinear"):
            if alg != "bicubic" and ow >= iw:
                continue
            plan = tts.Plan.from_vf("scale=%d:%d:flags=%s" % (ow, oh, alg), iw, ih)
            n = max(8, min(frames, int(3e9 // (plan.in_bytes + plan.out_bytes))))
            src = torch.randint(0, 256, (n, plan.in_bytes), dtype=torch.uint8, device="cuda")
            dst = torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda")
            for _ in range(5): plan.run_batch(src, dst)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): plan.run_batch(src, dst)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("%-11s %-11s %-9s %-30s %8.3f %6.1f" % ("%dx%d" % (iw, ih), "%dx%d" % (ow, oh), alg, plan.kernel_name(0), ms,
                                                        (plan.in_bytes + plan.out_bytes) * n / ms / 1e6 / 80), flush=True)
            del src, dst
