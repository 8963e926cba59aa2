#!/usr/bin/env python3
"""CPU baseline of every BASELINE config: the repo's own oracle (oracle/swscale_oracle.c, NOT ffmpeg), one thread
and T threads (frames are independent; ctypes releases the GIL).  Usage: python tests/tools/cpu_baseline.py [threads] [seconds]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from oracle import oracle
import tiktoken_amd as tts

PORTRAIT = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
JOBS = [("1080p->portrait", PORTRAIT, 1920, 1080), ("thumbnails", "scale=320:568:flags=area", 1080, 1920),
        ("4k->portrait", PORTRAIT, 3840, 2160), ("720p->portrait", PORTRAIT, 1280, 720)]


def rate(geom, src, threads, budget):
    def worker(k):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            oracle.run_chain(geom, src[(n + k) % len(src):][:1])
            n += 1
        return n
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        total = sum(ex.map(worker, range(threads)))
    return total / (time.perf_counter() - t0)


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    out = {"kind": "port (own restatement of the swscale C path, not ffmpeg)", "threads": threads, "host_cpus_visible": os.cpu_count()}
    for name, vf, w, h in JOBS:
        geom = tts.parse_vf(vf, w, h).as_dict()
        src = synth.frames("random", w, h, 2)
        oracle.run_chain(geom, src[:1])
        out[name] = {"fps_1_thread": round(rate(geom, src, 1, budget), 2), "fps_%d_threads" % threads: round(rate(geom, src, threads, budget), 2)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
