#!/usr/bin/env python3
"""Debug helper: one chain through tts_run (every plane its own allocation, tight strides); where does it differ from the oracle?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tiktoken_amd as tts, synth
from oracle import oracle
vf, iw, ih, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
plan = tts.Plan.from_vf(vf, iw, ih)
print("kinds", plan.kernel_kind(0), plan.kernel_kind(1))
ow, oh = plan.out_size
src = synth.frames("random", iw, ih, n)
want = oracle.run_chain(plan.geom, src)
sp, dp, ss, ds, keep = [], [], [], [], []
for i in range(n):
    planes, strides, outs, ostr = [], [], [], []
    for k, pl in enumerate(synth.split(src[i], iw, ih)):
        t = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
        keep.append(t); planes.append(t.data_ptr()); strides.append(pl.shape[1])
    for k in range(3):
        w_, h_ = (ow, oh) if k == 0 else ((ow + 1) // 2, (oh + 1) // 2)
        t = torch.full((h_, w_), 7, dtype=torch.uint8, device="cuda")
        keep.append(t); outs.append(t); ostr.append(w_)
    sp.append(planes); ss.append(strides); dp.append(outs); ds.append(ostr)
    print("frame", i, "src ptrs", [hex(p) for p in planes])
plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
torch.cuda.synchronize()
for i in range(n):
    for k, w in enumerate(synth.split(want[i], ow, oh)):
        g = dp[i][k].cpu().numpy()
        bad = g != w
        if bad.any():
            ys, xs = np.nonzero(bad)
            print("frame %d plane %d: %d bad, rows %d..%d cols %d..%d" % (i, k, bad.sum(), ys.min(), ys.max(), xs.min(), xs.max()))
