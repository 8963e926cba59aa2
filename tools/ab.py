#!/usr/bin/env python3
"""A/B of two builds of libttscale on the same box: python tools/ab.py <other.so> [case] [frames].
Interleaved rounds, median ms per batch."""
import ctypes, os, sys, subprocess, json
case = sys.argv[2] if len(sys.argv) > 2 else "portrait"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
if len(sys.argv) > 4:      # child: time one library
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tiktoken_amd._lib as L
    if sys.argv[4] != "-": L.LIB_PATH = sys.argv[4]
    import torch, tiktoken_amd as tts
    from sweep import CASES, IN_FMT
    vf, iw, ih, algo = CASES[case]
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=IN_FMT.get(case, "yuv420p"))
    src = torch.randint(0, 256, (frames, plan.in_bytes), dtype=torch.uint8, device="cuda")
    dst = torch.empty((frames, plan.out_bytes), dtype=torch.uint8, device="cuda")
    for _ in range(30): plan.run_batch(src, dst)
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.run_batch(src, dst)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    ts.sort(); print(json.dumps({"lib": sys.argv[4], "median_ms": ts[4], "min_ms": ts[0], "frac": algo * frames / ts[4] / 1e6 / 8000}))
else:
    for rnd in range(3):
        for lib in ("-", os.path.abspath(sys.argv[1])):
            print(subprocess.run([sys.executable, __file__, sys.argv[1], case, str(frames), lib], capture_output=True, text=True).stdout.strip(), flush=True)
