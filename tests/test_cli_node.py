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
    # the input's pixel format (`-pix_fmt rgb24` in front of `-i`) as an optional last argument
    js = "const t=require(%r);console.log(JSON.stringify(t.parseVf('scale=96:54',64,36,'rgb24')))" % os.path.join(ROOT, "js")
    out = subprocess.run([NODE, "-e", js], capture_output=True, text=True, check=True).stdout
    assert json.loads(out) == tts.parse_vf("scale=96:54", 64, 36, in_fmt="rgb24").as_dict()
    js = "const t=require(%r);try{t.parseVf('scale=96:54',64,36,'nv12')}catch(e){console.log(e.message)}" % os.path.join(ROOT, "js")
    assert "INVALID_INPUT" in subprocess.run([NODE, "-e", js], capture_output=True, text=True, check=True).stdout


@pytest.mark.gpu
@needs_node
def test_node_addon_rgb24_in_and_out(tts, oracle, tmp_path):
    """runHost with packed RGB on either side: `format=rgb24` ending the chain (the thumbnail's colour conversion,
    src/services/video.js:329-334) and an rgb24 source (`-pix_fmt rgb24` in front of `-i`)."""
    yuv = synth.frames("random", 64, 48, 2)
    rgb = np.random.default_rng(3).integers(0, 256, (2, 64 * 48 * 3), dtype=np.uint8)
    (tmp_path / "in.yuv").write_bytes(yuv.tobytes())
    (tmp_path / "in.rgb").write_bytes(rgb.tobytes())
    js = """
const fs = require('fs'), t = require(%r);
fs.writeFileSync(%r, t.runHost('format=rgb24', 64, 48, fs.readFileSync(%r)));
fs.writeFileSync(%r, t.runHost('scale=96:72', 64, 48, fs.readFileSync(%r), 'rgb24'));
""" % (os.path.join(ROOT, "js"), str(tmp_path / "out.rgb"), str(tmp_path / "in.yuv"), str(tmp_path / "out.yuv"), str(tmp_path / "in.rgb"))
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.frombuffer((tmp_path / "out.rgb").read_bytes(), np.uint8).reshape(2, -1)
    assert np.array_equal(got, oracle.run_chain(tts.parse_vf("format=rgb24", 64, 48), yuv))
    got = np.frombuffer((tmp_path / "out.yuv").read_bytes(), np.uint8).reshape(2, -1)
    assert np.array_equal(got, oracle.run_chain(tts.parse_vf("scale=96:72", 64, 48, in_fmt="rgb24"), rgb))


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


@needs_node
def test_video_processor_contract_without_gpu(tts):
    """web/src/tests/video/compression.test.js:52-90 and portrait.test.js:70-97,133-147: error names,
    9:16 -> `scale=720:1280`, anything else -> the `min()` / `decrease` chain."""
    js = """
const {videoProcessor: v, VIDEO_CONSTANTS: C} = require(%r);
const out = {};
out.portrait = v.scaleFilter({width: 1080, height: 1920});
out.square = v.scaleFilter({width: 1080, height: 1080});
out.tall = v.scaleFilter({width: 720, height: 2560});
const small = {type: 'video/mp4', width: 8, height: 8, data: Buffer.alloc(96)};
Promise.all([
  v.compress(null).catch((e) => e.message),
  v.compress(undefined).catch((e) => e.message),
  v.compress({type: 'video/webm', width: 8, height: 8, data: Buffer.alloc(96)}).catch((e) => e.message),
  v.compress({type: 'video/mp4', width: 8, height: 8, data: Buffer.alloc(95)}).catch((e) => e.message),
  v.validatePortraitMode({width: 720, height: 1280, size: 10}).then((r) => r.isValid),
  v.validatePortraitMode({size: 10}).catch((e) => e.message),                       // portrait.test.js:148-154
  v.validatePortraitMode({width: 720, height: 1280, size: 0}).catch((e) => e.message),   // portrait.test.js:127-131
  v.validatePortraitMode({width: 720, height: 1280, data: Buffer.alloc(0)}).catch((e) => e.message),
  v.compress(small).then((r) => r === small),                                       // compression.test.js:44-50: under 100 MB
  v.compress(Object.assign({size: 50 * 1024 * 1024}, small)).then((r) => r.size === 50 * 1024 * 1024 && r.data === small.data),
]).then((r) => {
  out.errs = r.slice(0, 4); out.valid = r[4]; out.meta = r[5]; out.empty = [r[6], r[7]]; out.gate = [r[8], r[9]];
  out.dropped = v.terminate();
  console.log(JSON.stringify(out));
});
""" % os.path.join(ROOT, "js", "videoProcessor.js")
    out = json.loads(subprocess.run([NODE, "-e", js], capture_output=True, text=True, check=True).stdout)
    assert out["portrait"] == "scale=720:1280"
    assert "force_original_aspect_ratio=decrease" in out["square"] and "min(720" in out["square"]
    assert "min(1280" in out["tall"]
    assert out["valid"] is True and out["meta"] == "Invalid video metadata"
    assert out["empty"] == ["Invalid video file", "Invalid video file"]
    assert out["gate"] == [True, True] and out["dropped"] == 0
    assert [e.split(":")[0] for e in out["errs"]] == ["INVALID_INPUT", "INVALID_INPUT", "INVALID_FORMAT", "INVALID_INPUT"]
    # the chains it emits resolve to what the web spec asks for
    assert tts.parse_vf(out["square"], 1080, 1080).as_dict()["out_w"] == 720
    g = tts.parse_vf(out["tall"], 720, 2560).as_dict()
    assert (g["out_w"], g["out_h"]) == (360, 1280)


@pytest.mark.gpu
@needs_node
def test_video_processor_compress_pixels(tts, oracle, tmp_path):
    src = synth.frames("random", 360, 640, 2)
    (tmp_path / "in.yuv").write_bytes(src.tobytes())
    js = """
const fs = require('fs');
const {videoProcessor: v} = require(%r);
const seen = [];
// (the spec's gate is the 150 MB container file; these raw frames stand for it)
v.compress({type: 'video/mp4', width: 360, height: 640, size: 150 * 1024 * 1024, data: fs.readFileSync(%r)}, {onProgress: (p) => seen.push(p)})
  .then((r) => { fs.writeFileSync(%r, r.data); console.log(JSON.stringify({w: r.width, h: r.height, vf: r.vf, seen, dropped: v.terminate()})); });
""" % (os.path.join(ROOT, "js", "videoProcessor.js"), str(tmp_path / "in.yuv"), str(tmp_path / "out.yuv"))
    r = json.loads(subprocess.run([NODE, "-e", js], capture_output=True, text=True, check=True).stdout)
    assert (r["w"], r["h"], r["vf"], r["seen"], r["dropped"]) == (720, 1280, "scale=720:1280", [0, 100], 1)
    got = np.frombuffer((tmp_path / "out.yuv").read_bytes(), np.uint8).reshape(2, -1)
    assert np.array_equal(got, oracle.run_chain(tts.parse_vf("scale=720:1280", 360, 640), src))


@needs_node
def test_loop_clips_driver(tts, oracle, tmp_path):
    """js/loopClips.js on a manifest shaped like VIDEOS_TO_EXTEND (scripts/loopVideos.js:11-34): the clip
    is replaced by initial + repeats x loop + end; no GPU involved (no pixel changes)."""
    clip = synth.frames("random", 32, 18, 30)
    (tmp_path / "birds_demo.yuv").write_bytes(clip.tobytes())
    manifest = {
        "birds_demo.yuv": {"width": 32, "height": 18, "fps": 5, "currentLength": 6, "targetLength": 10,
                           "loopStrategy": {"extractStart": 3, "extractDuration": 2, "repeats": 3, "crossfade": 0.5}},
        "missing.yuv": {"width": 32, "height": 18, "loopStrategy": {"extractStart": 0, "extractDuration": 1, "repeats": 2}},
    }
    (tmp_path / "manifest.json").write_text(json.dumps(manifest))
    r = subprocess.run([NODE, os.path.join(ROOT, "js", "loopClips.js"), str(tmp_path / "manifest.json")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rep = json.loads(r.stdout)
    assert rep["birds_demo.yuv"] == {"frames": 50, "duration": "10.000000"} and "error" in rep["missing.yuv"]
    got = np.frombuffer((tmp_path / "birds_demo.yuv").read_bytes(), np.uint8).reshape(50, -1)
    assert np.array_equal(got, clip[oracle.loop_order(30, 5, 3, 2, 3)])
    assert not (tmp_path / "extended_birds_demo.yuv").exists()


@needs_node
def test_shim_segment_extraction(tts, oracle, tmp_path):
    """ffmpeg(path).seekInput(start).duration(d).output(out).on('end').on('error').run(), the call chain of
    extractLoopSegment (scripts/loopVideos.js:42-52), on a raw clip at an NTSC rate."""
    clip = synth.frames("random", 32, 18, 60)
    (tmp_path / "in.yuv").write_bytes(clip.tobytes())
    js = """
const ffmpeg = require(%r);
ffmpeg(%r).inputSize(32, 18).inputFPS(29.97).seekInput(0.5).duration(1)
  .output(%r).on('end', () => console.log('END')).on('error', (e) => { console.log('ERR ' + e.message); process.exit(1); }).run();
""" % (os.path.join(ROOT, "js"), str(tmp_path / "in.yuv"), str(tmp_path / "seg.yuv"))
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
    assert r.returncode == 0 and "END" in r.stdout, r.stdout + r.stderr
    want = oracle.segment_frames(60, 30000 / 1001, 0.5, 1)
    assert want == list(range(15, 45))
    got = np.frombuffer((tmp_path / "seg.yuv").read_bytes(), np.uint8).reshape(len(want), -1)
    assert np.array_equal(got, clip[want])


@pytest.mark.gpu
@needs_node
def test_process_clips_batch_driver(tts, oracle, tmp_path):
    """js/processClips.js on a two-clip manifest shaped like VIDEOS_TO_FIX (scripts/processVideos.js:25-48):
    a small clip is resized to 720x1280 and trimmed, a large one is only trimmed; the report has the
    reference's {codec, resolution, duration} fields (:141-145)."""
    small = synth.frames("random", 360, 640, 6)
    large = synth.frames("zone", 720, 1280, 4)
    (tmp_path / "small.yuv").write_bytes(small.tobytes())
    (tmp_path / "large.yuv").write_bytes(large.tobytes())
    manifest = {
        "small.yuv": {"width": 360, "height": 640, "fps": 2, "issues": ["resolution", "duration"], "targetDuration": 2,
                      "trimStart": 1, "newName": "forest_demo.yuv"},
        "large.yuv": {"width": 720, "height": 1280, "fps": 2, "issues": ["resolution", "duration"], "targetDuration": 1,
                      "newName": "birds_demo.yuv"},
        "missing.yuv": {"width": 8, "height": 8, "issues": []},
    }
    (tmp_path / "manifest.json").write_text(json.dumps(manifest))
    r = subprocess.run([NODE, os.path.join(ROOT, "js", "processClips.js"), str(tmp_path / "manifest.json")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rep = json.loads(r.stdout)
    assert rep["small.yuv"] == {"codec": "rawvideo", "resolution": "720x1280", "duration": "2.000000"}
    assert rep["large.yuv"] == {"codec": "rawvideo", "resolution": "720x1280", "duration": "1.000000"}
    assert "error" in rep["missing.yuv"]
    got = np.frombuffer((tmp_path / "processed" / "forest_demo.yuv").read_bytes(), np.uint8).reshape(4, -1)
    assert np.array_equal(got, oracle.run_chain(tts.parse_vf("scale=w=720:h=1280", 360, 640), small[2:6]))
    got = np.frombuffer((tmp_path / "processed" / "birds_demo.yuv").read_bytes(), np.uint8).reshape(2, -1)
    assert np.array_equal(got, large[:2])
