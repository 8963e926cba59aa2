"""Host logic of libttscale that needs no GPU: the library loads and exports every symbol the header
declares, the filter-string parser resolves the reference's chains like libavfilter would, and the
product's own integer tables equal the oracle's."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(tts):
    header = open(os.path.join(ROOT, "include", "ttscale.h")).read()
    declared = set(re.findall(r"\b(tts_[a-z_]+)\s*\(", header))
    assert declared, "no declarations found in include/ttscale.h"
    handle = ctypes.CDLL(tts.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), "libttscale.so does not export %s" % name
    assert declared == set(tts.SYMBOLS), "python binding and header disagree: %s" % (declared ^ set(tts.SYMBOLS))
    assert tts._lib.lib().tts_version().decode().startswith("ttscale")


def test_missing_library_fails_loudly(tts, monkeypatch):
    monkeypatch.setattr(tts._lib, "_lib", None)
    monkeypatch.setattr(tts._lib, "LIB_PATH", "/nonexistent/libttscale.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tts._lib.lib()


# (vf, iw, ih) -> expected fields.  First three rows are the reference's own strings.
PARSE_CASES = [
    ("scale=w=720:h=1280", 480, 854,                                       # scripts/processVideos.js:96 via fluent-ffmpeg
     dict(do_scale=1, alg=0, sc_w=720, sc_h=1280, out_w=720, out_h=1280, do_pad=0)),
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 720, 1280,
     dict(sc_w=101, sc_h=180, do_pad=1, pad_x=108, pad_y=0, out_w=320, out_h=180, fill_y=16, fill_u=128, fill_v=128)),
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1920, 1080,
     dict(sc_w=320, sc_h=180, do_pad=1, pad_x=0, pad_y=0, out_w=320, out_h=180)),
    ("scale='min(720,iw)':'min(1280,ih)':force_original_aspect_ratio=decrease", 720, 2560,   # portrait.test.js:133-147
     dict(sc_w=360, sc_h=1280, out_w=360, out_h=1280)),
    ("scale=min(720\\,iw):min(1280\\,ih):force_original_aspect_ratio=decrease", 1080, 1080,  # portrait.test.js:85-97
     dict(sc_w=720, sc_h=720)),
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1920, 1080,      # BASELINE.json configs[1]
     dict(sc_w=3413, sc_h=1920, crop_x=1166, crop_y=0, crop_w=1080, crop_h=1920, out_w=1080, out_h=1920)),
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 3840, 2160,
     dict(sc_w=3413, sc_h=1920, crop_x=1166)),
    ("scale=320:568:flags=area", 1080, 1920, dict(alg=1, sc_w=320, sc_h=568)),
    ("scale=-2:360:flags=bilinear", 1280, 720, dict(alg=2, sc_w=640, sc_h=360)),
    ("scale=iw/2:ih/2", 1280, 720, dict(sc_w=640, sc_h=360)),
    ("crop=64:32:10:6,scale=128:96", 100, 60, dict(pre_x=10, pre_y=6, pre_w=64, pre_h=32, sc_w=128, sc_h=96)),
    ("crop=in_w-21:in_h", 101, 60, dict(pre_w=80, pre_h=60, pre_x=10, out_w=80)),       # 10.5 -> lrint 10 -> even
    ("pad=640:480:-1:-1:color=white", 320, 240, dict(do_pad=1, pad_x=160, pad_y=120, fill_y=235, fill_u=128, fill_v=128)),
    ("format=yuv420p", 64, 48, dict(do_scale=0, do_pad=0, out_w=64, out_h=48)),
    ("scale=64:48", 64, 48, dict(do_scale=0, out_range=0)),
    # video.js:329-334 writes a .jpg: the mjpeg encoder wants yuvj420p, i.e. one more (range) pass
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p,format=yuvj420p", 720, 1280,
     dict(sc_w=101, sc_h=180, do_pad=1, pad_x=108, out_range=1)),
    ("format=pix_fmts=yuvj420p", 64, 48, dict(do_scale=0, out_range=1)),
    # the scale itself writes yuvj420p unless a format=yuv420p pins the link behind it: range step inside its context, pad colour in full range
    ("scale=32:24,pad=40:30:-1:-1:color=red,format=yuvj420p", 64, 48, dict(do_scale=1, out_range=0, sc_range=1, rgb_stage=0, fill_y=76, fill_u=85, fill_v=255)),
    ("scale=32:24,pad=40:30:-1:-1:color=red,format=yuv420p,format=yuvj420p", 64, 48, dict(do_scale=1, out_range=1, sc_range=0, rgb_stage=0, fill_y=81, fill_u=90, fill_v=240)),
    ("scale=64:48,format=yuvj420p", 64, 48, dict(do_scale=0, out_range=1, sc_range=0, rgb_stage=0)),
    ("scale=32:24,format=yuvj420p,format=yuv420p", 64, 48, dict(do_scale=1, sc_range=1, out_range=2)),   # both steps may be set
    # ff_scale_adjust_dimensions: the aspect-derived size is rounded to the NEAREST multiple of force_divisible_by
    # first (101.25 -> 102), then min'ed and rounded down
    ("scale=320:180:force_original_aspect_ratio=decrease:force_divisible_by=2", 720, 1280, dict(sc_w=102, sc_h=180)),
    ("scale=320:180:force_original_aspect_ratio=decrease:force_divisible_by=4", 720, 1280, dict(sc_w=100, sc_h=180)),
    ("scale=1080:1920:force_original_aspect_ratio=increase:force_divisible_by=2", 1920, 1080, dict(sc_w=3414, sc_h=1920)),
    # packed RGB ending the chain: the chain's scale writes it, and crop / pad behind it work on
    # RGB pixels -- no chroma grid: the 101-wide picture sits at x = 109, not 108
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=rgb24", 720, 1280,
     dict(sc_w=101, sc_h=180, do_pad=1, pad_x=109, pad_y=0, out_w=320, out_h=180, out_fmt=1, rgb_stage=1, fill_rgb=0)),
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=rgb24", 1920, 1080,
     dict(sc_w=3413, crop_x=1166, out_fmt=1, rgb_stage=1)),
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1081:1919,format=bgra", 1920, 1080,   # odd sizes survive on RGB
     dict(crop_x=1166, crop_y=0, crop_w=1081, crop_h=1919, out_fmt=4, rgb_stage=1)),
    # a format=yuv420p between the scale and the RGB format pins that link: the conversion is a pass of its own at the end
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p,format=rgb24", 720, 1280,
     dict(pad_x=108, out_fmt=1, rgb_stage=2)),
    ("format=rgb24", 64, 48, dict(do_scale=0, out_fmt=1, rgb_stage=2)),
    ("crop=32:20,pad=64:48:-1:-1:color=white,format=pix_fmts=bgr24", 64, 48, dict(pre_w=32, pad_x=16, pad_y=14, out_fmt=2, rgb_stage=2, fill_rgb=0xffffff)),
]


@pytest.mark.parametrize("vf,iw,ih,want", PARSE_CASES)
def test_parse_vf(tts, vf, iw, ih, want):
    g = tts.parse_vf(vf, iw, ih).as_dict()
    for k, v in want.items():
        assert g[k] == v, "%s: %s = %d, want %d" % (vf, k, g[k], v)


@pytest.mark.parametrize("vf,name", [
    ("hflip", "INVALID_FORMAT"), ("scale=abc:def", "INVALID_FORMAT"), ("scale=10:10:flags=lanczos", "UNSUPPORTED"),
    ("format=gray", "UNSUPPORTED"), ("format=rgb24,crop=10:10", "UNSUPPORTED"), ("format=yuvj420p,scale=64:36,format=rgb24", "UNSUPPORTED"), ("format=yuvj420p,scale=64:36", "UNSUPPORTED"), ("scale=64:36,format=yuvj420p,format=rgb24", "UNSUPPORTED"),
    ("crop=4000:10", "INVALID_INPUT"), ("pad=10:10", "INVALID_INPUT"),
    ("scale=64:64,scale=32:32", "UNSUPPORTED"), ("scale=100:100:unknown=1", "INVALID_FORMAT"),
])
def test_parse_errors_carry_the_web_spec_names(tts, vf, name):
    # web/src/tests/video/compression.test.js:62-90 expects these words in thrown messages
    with pytest.raises(tts.TtsError, match=name):
        tts.parse_vf(vf, 640, 360)
    with pytest.raises(tts.TtsError, match="INVALID_INPUT"):
        tts.parse_vf("scale=10:10", 0, 10)


def test_parser_agrees_with_oracle_geometry(tts, oracle):
    for iw, ih, w, h, foar in [(1920, 1080, 1080, 1920, 2), (1280, 720, 1080, 1920, 2), (720, 1280, 320, 180, 1),
                               (1080, 1080, 320, 180, 1), (853, 481, 333, 777, 1), (853, 481, 333, 777, 2)]:
        word = {1: "decrease", 2: "increase"}[foar]
        g = tts.parse_vf("scale=%d:%d:force_original_aspect_ratio=%s" % (w, h, word), iw, ih)
        assert (g.sc_w, g.sc_h) == oracle.scale_dims(iw, ih, w, h, foar)
        for fdb in (2, 3, 4, 16):
            g = tts.parse_vf("scale=%d:%d:force_original_aspect_ratio=%s:force_divisible_by=%d" % (w, h, word, fdb), iw, ih)
            assert (g.sc_w, g.sc_h) == oracle.scale_dims(iw, ih, w, h, foar, fdb)
    for inn, out in [(3413, 1080), (1920, 1080), (101, 100), (3413, 1082)]:
        g = tts.parse_vf("crop=%d:2" % out, inn, 4)          # a crop with no scale before it is a source crop
        assert g.pre_x == oracle.crop_offset(inn, out)


@pytest.mark.parametrize("src,dst", [(1920, 3413), (1080, 1920), (960, 1707), (540, 960), (3840, 3413), (2160, 1920),
                                     (1280, 3413), (720, 1920), (720, 101), (1280, 180), (1080, 320), (1920, 568),
                                     (100, 100), (101, 100), (7, 3), (3, 7), (2, 5), (1, 4), (4, 1), (5, 64)])
def test_product_tables_equal_oracle_tables(tts, oracle, src, dst):
    for alg in (0, 1, 2):
        for one in (1 << 14, 1 << 12):
            a, b = oracle.filter_table(src, dst, one, alg), tts.build_filter(src, dst, one, alg)
            assert a[0].shape == b[0].shape and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_parse_vf_rgb24_input(tts):
    # a crop in front of the scale works on RGB pixels (no chroma grid); behind it the planes are yuv420p again
    g = tts.parse_vf("crop=64:33:11:7,scale=128:96,pad=140:100:-1:-1", 100, 60, in_fmt="rgb24").as_dict()
    assert (g["in_fmt"], g["pre_x"], g["pre_y"], g["pre_w"], g["pre_h"]) == (1, 11, 7, 64, 33)
    assert (g["pad_x"], g["pad_y"], g["out_w"], g["out_h"], g["out_fmt"]) == (6, 2, 140, 100, 0)
    # no scale: the pad works on RGB pixels (any offset, odd picture sizes) in front of the conversion pass: rgb_stage 2
    g = tts.parse_vf("crop=31:20:3:1,pad=80:61:-1:-1:color=red", 64, 36, in_fmt="bgr24").as_dict()
    assert (g["rgb_stage"], g["pre_x"], g["pre_w"], g["pad_x"], g["pad_y"], g["out_w"], g["out_h"], g["fill_rgb"]) == (2, 3, 31, 24, 20, 80, 61, 0xff0000)
    with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
        tts.parse_vf("pad=81:60", 64, 36, in_fmt="rgb24")               # an odd canvas into the pair-summing chroma reader
    assert tts.parse_vf("scale=64:36,pad=80:60", 64, 36, in_fmt="rgb24").as_dict()["rgb_stage"] == 0
    # a format=yuv420p in front of everything else is where the conversion pass goes: behind it the chain sees yuv420p
    g = tts.parse_vf("crop=32:20:3:1,format=yuv420p,crop=16:10:2:2,pad=40:30", 64, 36, in_fmt="rgb24").as_dict()
    assert (g["pre_x"], g["pre_y"], g["pre_w"], g["do_scale"], g["crop_x"], g["crop_y"], g["crop_w"], g["do_pad"], g["rgb_stage"]) == (3, 1, 32, 0, 2, 2, 16, 1, 0)
    with pytest.raises(tts.TtsError, match="two swscale contexts"):
        tts.parse_vf("format=yuv420p,scale=96:54", 64, 36, in_fmt="rgb24")
    with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
        tts.parse_vf("scale=10:10,format=rgb24", 64, 36, in_fmt="rgb24")
    with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
        tts.parse_vf("crop=31:20", 64, 36, in_fmt="rgb24")              # odd width into the pair-averaging chroma reader
    with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
        tts.parse_vf("crop=31:20,scale=24:20", 64, 36, in_fmt="rgb24")  # ... a shrink reads pairs too
    # a widening reads chroma per pixel (sws_init_context: chrSrcHSubSample stays 0): an odd width is fine there
    assert tts.parse_vf("crop=31:20,scale=48:30", 64, 36, in_fmt="rgb24").as_dict()["pre_w"] == 31


def test_parse_vf_rgb_from_a_full_range_link(tts):
    # `format=yuvj420p,format=rgb24`: the range pass stays in out_range, the conversion pass sits at the end
    g = tts.parse_vf("crop=32:20,format=yuvj420p,format=bgra", 64, 48).as_dict()
    assert (g["out_range"], g["out_fmt"], g["rgb_stage"], g["out_w"], g["out_h"]) == (1, 4, 2, 32, 20)


def test_frame_bytes(tts):
    assert tts.frame_bytes(1920, 1080) == 3110400
    assert tts.frame_bytes(101, 57) == 101 * 57 + 2 * 51 * 29


def test_sizes_out_of_any_frames_range_are_refused(tts):
    """Found by tools/hostfuzz (ASan + UBSan over the parser): `scale=-2147483648:...` negated INT_MIN, and a pad
    canvas had no upper limit.  Both are plain INVALID_INPUT now."""
    for vf in ("scale=-2147483648:180", "scale=320:-2147483648", "scale=99999999999:180",
               "scale=320:180,pad=332520:180:-1:-1", "scale=320:180,pad=320:70000"):
        with pytest.raises(tts.TtsError):
            tts.parse_vf(vf, 1920, 1080)


def test_host_fuzz_under_sanitizers():
    """tools/hostfuzz/run.sh: 20,000 mutated filter strings through parse_vf / check_geom / build_axis_filter in a
    g++ -fsanitize=address,undefined build (host code only; the GPU pool offers no device sanitizer)."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "tools", "hostfuzz", "run.sh"), "20000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no finding" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_parse_vf_full_range_source(tts):
    """`-pix_fmt yuvj420p -i`: in_fmt stays 0 (the planes are the same); every link has a range and the steps sit where it changes."""
    T = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=red,format=yuv420p"
    for vf, want in (("scale=w=720:h=1280", dict(sc_range=0, out_range=0)),
                     (T, dict(sc_range=2, out_range=0, fill_y=81, fill_u=90, fill_v=240)),             # the pad behind the scale: on the limited link
                     (T + ",format=yuvj420p", dict(sc_range=2, out_range=1, fill_y=81)),
                     ("scale=320:180,pad=400:200:-1:-1:color=red", dict(sc_range=0, out_range=0, fill_y=76, fill_u=85, fill_v=255)),   # on a full-range link
                     ("scale=320:180,format=yuvj420p", dict(sc_range=0, out_range=0)),
                     ("format=yuv420p", dict(sc_range=0, out_range=2, do_scale=0))):
        g = tts.parse_vf(vf, 640, 360, in_fmt="yuvj420p").as_dict()
        assert g["in_fmt"] == 0
        for k, v in want.items():
            assert g[k] == v, (vf, k, g[k], v)
    for vf in ("scale=320:180,format=rgb24", "format=yuv420p,scale=320:180", "format=rgb24"):
        with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
            tts.parse_vf(vf, 640, 360, in_fmt="yuvj420p")
    with pytest.raises(tts.TtsError, match="two range passes"):
        tts.parse_vf("format=yuvj420p,format=yuv420p", 640, 360)


def test_parse_vf_packed_rgb_source_range_steps(tts):
    """Fed with packed RGB: the first yuv format is what the converting context writes (sc_range 1 for yuvj420p), one of the other
    range behind it is a pass of its own at the end (out_range) -- the reference's thumbnail chain + its mjpeg encoder."""
    T = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=red"
    for vf, want in ((T + ",format=yuv420p,format=yuvj420p", dict(sc_range=0, out_range=1, rgb_stage=0, fill_y=81, fill_u=90, fill_v=240)),
                     (T + ",format=yuvj420p", dict(sc_range=1, out_range=0, fill_y=76, fill_u=85, fill_v=255)),
                     (T + ",format=yuvj420p,format=yuv420p", dict(sc_range=1, out_range=2, fill_y=76)),
                     ("format=yuv420p,format=yuvj420p", dict(sc_range=0, out_range=1, do_scale=0)),
                     ("pad=700:400:-1:-1:color=red,format=yuv420p,format=yuvj420p", dict(sc_range=0, out_range=1, rgb_stage=2)),
                     ("format=yuv420p,pad=700:400:-1:-1:color=red,format=yuvj420p", dict(sc_range=0, out_range=1, rgb_stage=0, fill_y=81)),
                     ("format=yuvj420p,pad=700:400:-1:-1:color=red", dict(sc_range=1, out_range=0, rgb_stage=0, fill_y=76)),
                     ("scale=320:180", dict(sc_range=0, out_range=0))):
        g = tts.parse_vf(vf, 640, 360, in_fmt="rgb24").as_dict()
        for k, v in want.items():
            assert g[k] == v, (vf, k, g[k], v)
    with pytest.raises(tts.TtsError, match="two range passes"):
        tts.parse_vf("scale=320:180,format=yuv420p,format=yuvj420p,format=yuv420p", 640, 360, in_fmt="rgb24")


def test_pad_must_contain_the_picture_after_rounding(tts):
    """vf_pad rounds its canvas DOWN to the chroma grid and then insists that the input area lies within it: an odd picture
    height that only fitted the unrounded canvas is an error in ffmpeg, and here (it used to be let through with a row to spare)."""
    with pytest.raises(tts.TtsError, match="INVALID"):
        tts.parse_vf("scale=24:43,pad=42:55:0:12:color=black", 242, 137)
    with pytest.raises(tts.TtsError, match="INVALID"):
        tts.parse_vf("scale=24:43,pad=42:55:0:12:color=black", 242, 137, in_fmt="bgr24")
    g = tts.parse_vf("scale=24:43,pad=42:56:0:12:color=black", 242, 137).as_dict()      # fits: all 43 rows are kept
    assert (g["pad_y"], g["crop_h"], g["out_h"]) == (12, 43, 56)
    g = tts.parse_vf("scale=101:180,pad=320x180:-1:-1", 1080, 1920).as_dict()            # an odd width that fits: kept in the geometry (its last column goes to the fill)
    assert (g["pad_x"], g["crop_w"], g["out_w"]) == (108, 101, 320)
