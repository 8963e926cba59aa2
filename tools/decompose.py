#!/usr/bin/env python3
"""Where a launch's time goes: the same instruction stream with the source and / or destination traffic dropped
(zero-record buffer descriptors) or kept in L2 (every frame at frame 0's address).  Outputs are wrong by design:
experiments build only (make -C tiktoken_amd/csrc exp; TTS_LIB=tiktoken_amd/libttscale_exp.so).
Usage: TTS_LIB=... python tools/decompose.py [case] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiktoken_amd as tts
from sweep import CASES, plan_for

case = sys.argv[1] if len(sys.argv) > 1 else "thumbref"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
vf, iw, ih, algo = CASES[case]
VARIANTS = [("as shipped", {}), ("no source traffic", {"TTS_EXP_NOSRC": "1"}), ("no destination traffic", {"TTS_EXP_NODST": "1"}),
            ("neither (instruction stream alone)", {"TTS_EXP_NOSRC": "1", "TTS_EXP_NODST": "1"}),
            ("source in L2 (every frame reads frame 0)", {"TTS_EXP_SRCPITCH0": "1"}),
            ("destination in L2", {"TTS_EXP_DSTPITCH0": "1"})]
plans = []
for name, env in VARIANTS:
    for k in ("TTS_EXP_NOSRC", "TTS_EXP_NODST", "TTS_EXP_SRCPITCH0", "TTS_EXP_DSTPITCH0"):
        os.environ.pop(k, None)
    os.environ.update(env)
    plans.append((name, plan_for(case)))
src = torch.randint(0, 256, (frames, plans[0][1].in_bytes), dtype=torch.uint8, device="cuda")
dst = torch.empty((frames, plans[0][1].out_bytes), dtype=torch.uint8, device="cuda")
for _ in range(30):
    plans[0][1].run_batch(src, dst)
best = {}
for rep in range(3):                                   # interleaved, so that clock drift hits every variant alike
    for name, plan in plans:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan.run_batch(src, dst)
        e0.record()
        for _ in range(20):
            plan.run_batch(src, dst)
        e1.record(); torch.cuda.synchronize()
        best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 20)
print("%s, %d frames, kernel %s" % (case, frames, plans[0][1].kernel_name(0)))
for name, _ in plans:
    print("  %-44s %8.1f us   %5.1f %% of the roofline if it were the whole job" % (name, best[name] * 1e3, 100 * frames * algo / (best[name] * 1e-3) / 8e12))
