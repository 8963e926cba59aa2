#!/usr/bin/env python3
"""Do consecutive clips overlap their launch ramps and tails when they alternate between two streams?
300-frame clips, 1 stream vs 2: python tools/twostream.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tiktoken_amd as tts
VF = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
for iw, ih in ((1280, 720), (1920, 1080), (3840, 2160)):
    plan = tts.Plan.from_vf(VF, iw, ih)
    n, clips = 300, 8
    src = [torch.randint(0, 256, (n, plan.in_bytes), dtype=torch.uint8, device="cuda") for _ in range(2)]
    dst = [torch.empty((n, plan.out_bytes), dtype=torch.uint8, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for nstreams in (1, 2):
        def burst():
            for c in range(clips):
                s = streams[c % nstreams]
                plan.run_batch(src[c % 2], dst[c % 2], stream=s.cuda_stream)
        for _ in range(5): burst()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for s in streams[:nstreams]: s.wait_event(e0)
            burst()
            for s in streams[:nstreams]: torch.cuda.current_stream().wait_stream(s)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / clips)
        ts.sort()
        print("%dx%d: %d stream(s): %.4f ms per 300-frame clip" % (iw, ih, nstreams, ts[3]), flush=True)
