#!/usr/bin/env python3
"""How long tts_plan_create takes (host table building + one small upload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiktoken_amd as tts
from sweep import CASES
tts.Plan.from_vf("scale=64:36", 128, 72)
for k, (vf, iw, ih, _) in CASES.items():
    t0 = time.perf_counter(); p = tts.Plan.from_vf(vf, iw, ih); dt = time.perf_counter() - t0
    print("%-18s %7.2f ms  %s" % (k, dt * 1e3, p.kernel_name(0)))
