#!/usr/bin/env python3
"""Rate of the per-frame pointer API (tts_run) against the packed batch API on the same frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tiktoken_amd as tts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
plan = tts.Plan.from_vf("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1920, 1080)
src = torch.randint(0, 256, (n, plan.in_bytes), dtype=torch.uint8, device="cuda")
dst = torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda")
iw, ih = plan.in_size; ow, oh = plan.out_size
def planes(t, w, h):
    base, fb = t.data_ptr(), t.shape[1]
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return [[base + i * fb, base + i * fb + w * h, base + i * fb + w * h + cw * ch] for i in range(n)], [[w, cw, cw]] * n
sp, ss = planes(src, iw, ih); dp, ds = planes(dst, ow, oh)
def timeit(fn):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 10
print("batch API   : %.4f ms per %d frames" % (timeit(lambda: plan.run_batch(src, dst)), n))
print("pointer API : %.4f ms per %d frames (includes building the ctypes frame arrays)" % (timeit(lambda: plan.run_frames(sp, dp, ss, ds)), n))
