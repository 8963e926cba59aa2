#!/usr/bin/env python3
"""Stress helper: one chain through tts_run many times with shuffled plane placement; report mismatches vs the oracle."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tiktoken_amd as tts, synth
from oracle import oracle
vf, iw, ih, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
plan = tts.Plan.from_vf(vf, iw, ih)
print("kinds", plan.kernel_kind(0), plan.kernel_kind(1))
ow, oh = plan.out_size
src = synth.frames("random", iw, ih, n)
want = oracle.run_chain(plan.geom, src)
rng = random.Random(1)
bad_runs = 0
for rep in range(reps):
    junk = [torch.full((rng.randrange(1, 3000) * 512,), 0xAB, dtype=torch.uint8, device="cuda") for _ in range(rng.randrange(0, 6))]
    sp, dp, ss, ds, keep = [], [], [], [], []
    order = list(range(n)); rng.shuffle(order)
    slots = {}
    for i in order:
        planes, strides, outs, ostr = [], [], [], []
        pls = synth.split(src[i], iw, ih)
        planes, strides = [0, 0, 0], [0, 0, 0]
        korder = [0, 1, 2]; rng.shuffle(korder)                 # any plane may end up highest in memory
        for k in korder:
            pl = pls[k]
            t = torch.zeros(pl.shape, dtype=torch.uint8, device="cuda")
            t[:] = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
            keep.append(t); planes[k] = t.data_ptr(); strides[k] = pl.shape[1]
            if rng.random() < 0.3: keep.append(torch.full((rng.randrange(1, 64) * 512,), 0xCD, dtype=torch.uint8, device="cuda"))
        for k in range(3):
            w_, h_ = (ow, oh) if k == 0 else ((ow + 1) // 2, (oh + 1) // 2)
            t = torch.full((h_, w_), 7, dtype=torch.uint8, device="cuda")
            keep.append(t); outs.append(t); ostr.append(w_)
        slots[i] = (planes, strides, outs, ostr)
    for i in range(n):
        a, b, c, d = slots[i]; sp.append(a); ss.append(b); dp.append(c); ds.append(d)
    plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        for k, w in enumerate(synth.split(want[i], ow, oh)):
            g = dp[i][k].cpu().numpy()
            bad = g != w
            if bad.any():
                ys, xs = np.nonzero(bad)
                allp = sorted((p, fi, kk) for fi in range(n) for kk, p in enumerate(sp[fi]))
                print("rep %d frame %d plane %d: %d bad, rows %d..%d cols %d..%d; got %s want %s; this plane @%x, highest plane: frame %d plane %d @%x" % (
                    rep, i, k, bad.sum(), ys.min(), ys.max(), xs.min(), xs.max(), g[bad][:6], w[bad][:6], sp[i][k], allp[-1][1], allp[-1][2], allp[-1][0]))
                bad_runs += 1
    del junk, keep
print("mismatching planes:", bad_runs)
