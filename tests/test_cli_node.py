"""The two non-Python front ends over the C ABI: the raw-YUV CLI (stand-in for the `ffmpeg`
executable) and the Node addon with its fluent-ffmpeg-shaped shim (scripts/processVideos.js:71-108)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tiktoken_amd", "ttscale")
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "js", "ttscale.node")
THUMB = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p"

needs_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or js/ttscale.node missing")


def test_cli_rejects_bad_chain_like_ffmpeg(tts):
    assert os.path.exists(CLI), "build tiktoken_amd/ttscale with make -C tiktoken_amd/csrc"
    r = subprocess.run([CLI, "-s", "64x36", "-i", "/dev/null", "-vf", "hflip", "/dev/null"], capture_output=True, text=True)
    assert r.returncode != 0 and "INVALID_FORMAT" in r.stderr
    r = subprocess.run([CLI, "-version"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ttscale")


@needs_node
def test_node_addon_parses_like_python(tts):
    js = "const t=require(%r);console.log(JSON.stringify(t.parseVf(%r,720,1280)))" % (os.path.join(ROOT, "js"), THUMB)
    out = subprocess.run([NODE, "-e", js], capture_output=True, text=True, check=True).stdout
    assert json.loads(out) == tts.parse_vf(THUMB, 720, 1280).as_dict()
    js = "const t=require(%r);try{t.parseVf('hflip',8,8)}catch(e){console.log(e.message)}" % os.path.join(ROOT, "js")
    assert "INVALID_FORMAT" in subprocess.run([NODE, "-e", js], capture_output=True, text=True, check=True).stdout


@pytest.mark.gpu
def test_cli_matches_oracle(tts, oracle, tmp_path):
    src = synth.frames("random", 480, 854, 3)
    (tmp_path / "in.yuv").write_bytes(src.tobytes())
    subprocess.run([CLI, "-f", "rawvideo", "-pix_fmt", "yuv420p", "-s", "480x854", "-i", str(tmp_path / "in.yuv"),
                    "-vf", "scale=w=720:h=1280", "-y", str(tmp_path / "out.yuv")], check=True)
    got = np.frombuffer((tmp_path / "out.yuv").read_bytes(), np.uint8).reshape(3, -1)
    assert np.array_equal(got, oracle.run_chain(tts.parse_vf("scale=w=720:h=1280", 480, 854), src))


@pytest.mark.gpu
@needs_node
def test_node_shim_runs_the_reference_call_chain(tts, oracle, tmp_path):
    """ffmpeg(input).outputOptions([...]).size('720x1280').on('end').on('error').save(out), as in
    scripts/processVideos.js:71-108, on a raw clip."""
    src = synth.frames("random", 480, 854, 2)
    (tmp_path / "in.yuv").write_bytes(src.tobytes())
    js = """
const ffmpeg = require(%r);
ffmpeg(%r).inputSize(480, 854)
  .outputOptions(['-c:v libx264', '-preset slow', '-crf 23', '-pix_fmt yuv420p'])
  .size('720x1280')
  .on('end', () => { console.log('END'); })
  .on('error', (e) => { console.log('ERR ' + e.message); process.exit(1); })
  .save(%r);
""" % (os.path.join(ROOT, "js"), str(tmp_path / "in.yuv"), str(tmp_path / "out.yuv"))
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
    assert r.returncode == 0 and "END" in r.stdout, r.stdout + r.stderr
    got = np.frombuffer((tmp_path / "out.yuv").read_bytes(), np.uint8).reshape(2, -1)
    assert np.array_equal(got, oracle.run_chain(tts.parse_vf("scale=w=720:h=1280", 480, 854), src))
