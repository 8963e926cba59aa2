#!/usr/bin/env python3
"""Debug helper: run one chain on the GPU and describe where it differs from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tiktoken_amd as tts, synth
from oracle import oracle
vf, iw, ih, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
plan = tts.Plan.from_vf(vf, iw, ih)
src = synth.frames("random", iw, ih, n)
want = oracle.run_chain(plan.geom, src)
got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
ow, oh = plan.out_size
for f in range(n):
    for name, g, w in zip("YUV", synth.split(got[f], ow, oh), synth.split(want[f], ow, oh)):
        bad = g != w
        if bad.any():
            cols = np.flatnonzero(bad.any(0)); rows = np.flatnonzero(bad.any(1))
            print("frame %d plane %s: %d bad; cols %s..%s (n=%d) first cols %s; rows %s..%s (n=%d)" % (
                f, name, bad.sum(), cols[0], cols[-1], cols.size, cols[:24].tolist(), rows[0], rows[-1], rows.size))
        else:
            print("frame %d plane %s ok" % (f, name))
if len(sys.argv) > 5:
    f = n - 1
    g, w = synth.split(got[f], ow, oh)[2], synth.split(want[f], ow, oh)[2]
    print("got V tail:\n", g[-6:, -10:]); print("want V tail:\n", w[-6:, -10:])
    s = synth.split(src[f], iw, ih)[2]
    print("src V tail:\n", s[-4:, -10:])
