#!/usr/bin/env python3
"""Write profiles/traffic.json from a tools/pmc.sh summary: the HBM bytes per frame of the hot kernel, stamped
with the hash of the kernel sources it was measured on (bench.py quotes it only while that hash matches).
Usage: python tools/traffic.py gpurun_out/<pmc dir>/summary.txt <frames per launch> [committed copy of the summary]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import ALGO_BYTES_PER_FRAME, kernel_sources_hash

summary, frames = sys.argv[1], int(sys.argv[2])
source = sys.argv[3] if len(sys.argv) > 3 else summary
vals, kernel = {}, None
for line in open(summary):
    if line.startswith("kernel "):
        kernel = line.split(None, 1)[1].strip()
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
    if m and kernel and kernel.startswith("strip_lds"):
        vals[m.group(1)] = float(m.group(2))
assert "FETCH_SIZE" in vals and "WRITE_SIZE" in vals, "summary holds no FETCH_SIZE / WRITE_SIZE of a strip_lds kernel"
# gfx950: FETCH_SIZE (KB) undercounts 16-B-per-lane reads by half (MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KB) as is
hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
out = {
    "kernel": kernel, "workload": "%d frames 1920x1080 -> portrait 1080x1920, one launch" % frames,
    "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of tools/pmc.sh, mean over dispatches)" % source,
    "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE doubled for 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
    "hbm_bytes_per_launch": int(hbm), "frames_per_launch": frames, "hbm_bytes_per_frame": hbm / frames,
    "algorithmic_bytes_per_frame": ALGO_BYTES_PER_FRAME, "ratio": round(hbm / frames / ALGO_BYTES_PER_FRAME, 4),
    "kernel_sources_sha256": kernel_sources_hash(),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
