"""GPU parity: libttscale (through the C ABI) vs the CPU oracle, bit-exact (0 LSB).

The oracle itself is PARITY UNPINNED against real FFmpeg (oracle/swscale_oracle.c header):
these tests prove HIP == oracle, not oracle == swscale."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

PORTRAIT = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
THUMB_REF = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p"

# (vf, iw, ih): the reference's own chains first, BASELINE.json's shapes, then edge geometries
CASES = [
    ("scale=w=720:h=1280", 480, 854),                 # scripts/processVideos.js:96 on a sub-720p clip
    ("scale=w=720:h=1280", 360, 640),
    (THUMB_REF, 720, 1280),                            # src/services/video.js:329-332, portrait source
    (THUMB_REF, 1280, 720),                            # landscape source (test/helpers/video.js)
    (THUMB_REF, 1080, 1080),                           # square (portrait.test.js:86)
    ("scale='min(720,iw)':'min(1280,ih)':force_original_aspect_ratio=decrease", 720, 2560),  # portrait.test.js:136-138
    (PORTRAIT, 1920, 1080),                            # BASELINE configs[1]
    (PORTRAIT, 1280, 720),
    (PORTRAIT, 3840, 2160),
    (PORTRAIT, 7680, 4320),                            # 8K: 10 taps per axis -> generic kernel
    ("scale=3840:2160", 1920, 1080),                   # 1080p -> 4K: the largest output here (12.4 MB a frame)
    ("scale=320:568:flags=area", 1080, 1920),          # BASELINE configs[2]
    ("scale=160:284:flags=area", 1080, 1920),
    ("scale=640:360:flags=bilinear", 1280, 720),
    ("scale=1281:721", 1280, 720),                     # barely scaled, odd output
    ("scale=101:57", 64, 36),                          # odd everything, tiny
    ("scale=64:36", 101, 57),
    ("scale=8:8", 6, 6),
    ("scale=2:2", 64, 64),                             # huge shrink: many taps -> generic kernel
    ("scale=500:2", 16, 300),
    ("crop=64:32:10:6,scale=128:96", 100, 60),         # crop before scale
    ("crop=320:180", 640, 360),                        # crop only
    ("pad=640:480:-1:-1:color=white", 320, 240),      # pad only
    ("scale=iw/2:-2,pad=iw+20:ih+10:10:4:color=red", 200, 150),
    ("format=yuv420p", 64, 48),                        # identity
    (THUMB_REF + ",format=yuvj420p", 720, 1280),       # video.js:329-334: the mjpeg encoder's full-range pass, borders included
    ("scale=320:180,format=yuvj420p", 1280, 720),      # the scale itself writes yuvj420p: its context converts the range between its two passes
    ("scale=320:180,format=yuv420p,format=yuvj420p", 1280, 720),   # pinned to yuv420p first (the reference's chain): a range pass of its own at the end
    ("scale=301:171:flags=area,crop=200:100:40:30,pad=240:128:-1:-1:color=red,format=yuvj420p", 1280, 720),   # ... crop and pad then work on yuvj420p: the pad colour in full range
    ("format=yuvj420p", 64, 48),                       # range step alone
]


def run_case(tts, oracle, vf, iw, ih, patterns, n=2):
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih)
    for pat in patterns:
        src = synth.frames(pat, iw, ih, n)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda())
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d %s: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, pat, bad.size, bad[0] if bad.size else -1,
            got.ravel()[bad[0]] if bad.size else 0, want.ravel()[bad[0]] if bad.size else 0)
    plan.close()


@pytest.mark.parametrize("vf,iw,ih", CASES)
def test_chain_matches_oracle(tts, oracle, vf, iw, ih):
    big = iw * ih > 1000 * 1000
    pats = ("random", "impulse") if big else synth.PATTERNS
    run_case(tts, oracle, vf, iw, ih, pats, n=1 if big else 3)


def test_batch_of_frames_ragged_tail(tts, oracle):
    # 37 frames: the last workgroup of every launch is partly idle
    run_case(tts, oracle, "scale=96:54", 64, 36, ("random",), n=37)


def test_empty_batch(tts):
    import torch
    plan = tts.Plan.from_vf("scale=96:54", 64, 36)
    out = plan.run_batch(torch.empty((0, plan.in_bytes), dtype=torch.uint8, device="cuda"))
    assert out.shape == (0, plan.out_bytes)


def test_generic_and_strip_kernels_agree(tts, oracle, monkeypatch):
    import torch
    src = synth.frames("random", 320, 180, 2)
    plan = tts.Plan.from_vf("scale=568:320", 320, 180)
    assert plan.kernel_kind(0) >= 1 and plan.kernel_kind(1) >= 1
    a = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    monkeypatch.setenv("TTS_FORCE_GENERIC", "1")
    plan2 = tts.Plan.from_vf("scale=568:320", 320, 180)
    assert plan2.kernel_kind(0) == 0
    b = plan2.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(a, oracle.run_chain(plan.geom, src))


@pytest.mark.parametrize("seg", [16, 33, 64, 4096])
def test_row_segmentation_is_invisible(tts, oracle, monkeypatch, seg):
    import torch
    monkeypatch.setenv("TTS_SEG_ROWS", str(seg))
    src = synth.frames("random", 200, 150, 2)
    for vf in ("scale=355:267", "scale=120:90:flags=area"):
        plan = tts.Plan.from_vf(vf, 200, 150)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        assert np.array_equal(got, oracle.run_chain(plan.geom, src))


@pytest.mark.parametrize("vf,iw,ih,n,pad", [
    ("scale=284:160", 160, 90, 5, True),
    ("scale=284:160", 160, 90, 130, False),      # more than one launch group (120 frames each), tight strides
    ("scale=136:96", 100, 60, 40, False),        # rows of 100 / 50 bytes: the runs of the LDS walk must stay inside each plane
    ("scale=72:40", 54, 30, 33, False),          # planes of a single short segment -> register walk in pointer mode
    ("scale=640:360:flags=area", 1283, 721, 3, False),
    # plane-by-plane launches (what tts_run falls back to when a frame's planes lie > 4 GiB apart), down
    # to ONE plane per launch: the plane then starts at offset 0 and ends the span
    ("scale=640:360:flags=area", 1283, 721, 1, "planewise"),
    ("scale=136:96", 100, 60, 1, "planewise"),
    ("scale=355:267", 200, 150, 2, "planewise"),
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1280, 720, 1, "planewise"),
    # the column walk in pointer mode
    (THUMB_REF, 1920, 1080, 3, False),
    (THUMB_REF, 1283, 721, 1, "planewise"),
    ("scale=213:121", 640, 360, 3, False),                  # ... two columns per lane
    ("scale=213:121", 640, 360, 1, "planewise"),
    ("scale=57:17", 324, 105, 2, False),                    # 26 vertical taps on 53 chroma rows: the first of two segments reaches the last row
    ("scale=20:12", 200, 120, 40, False),                   # 12-row planes: single segment -> generic kernel in pointer mode
    # source planes too small for a staged run to be slid inside them: 24 / 12-byte rows, 6 / 3-row planes
    ("scale=136:96", 24, 60, 33, False),
    ("scale=320:200", 200, 6, 33, False),
    ("scale=20:4", 200, 6, 33, False),
    # the range step inside the scale's context (tts_geom.sc_range) in pointer mode: the RNG strip and column kernels, and the
    # general kernel where pointer mode would take the register walk (which has no such form)
    ("scale=284:160,format=yuvj420p", 160, 90, 5, True),
    ("scale=284:160,format=yuvj420p", 160, 90, 130, False),
    ("scale=72:40,format=yuvj420p", 54, 30, 33, False),
    ("scale=136:96,format=yuvj420p", 24, 60, 33, False),
    ("scale=355:267,format=yuvj420p", 200, 150, 2, "planewise"),
    ("scale=213:121,format=yuvj420p", 640, 360, 3, False),
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=yuvj420p", 1280, 720, 1, "planewise"),
])
def test_frame_pointer_api_matches_batch_api(tts, oracle, monkeypatch, vf, iw, ih, n, pad):
    """tts_run with per-frame plane pointers (every plane its own allocation, padded or tight strides)
    == the oracle."""
    import torch
    if pad == "planewise":
        monkeypatch.setenv("TTS_PLANEWISE", "1")
        pad = False
    plan = tts.Plan.from_vf(vf, iw, ih)
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    sp, dp, ss, ds, keep = [], [], [], [], []
    for i in range(n):
        y, u, v = synth.split(src[i], iw, ih)
        planes, strides = [], []
        for k, pl in enumerate((y, u, v)):
            stride = pl.shape[1] + ((24 if k == 0 else 8) if pad else 0)
            t = torch.zeros((pl.shape[0], stride), dtype=torch.uint8, device="cuda")
            t[:, : pl.shape[1]] = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
            keep.append(t); planes.append(t.data_ptr()); strides.append(stride)
        sp.append(planes); ss.append(strides)
        outs, ostr = [], []
        for k in range(3):
            w_, h_ = (ow, oh) if k == 0 else ((ow + 1) // 2, (oh + 1) // 2)
            stride = w_ + (16 if pad else 0)
            t = torch.full((h_, stride), 7, dtype=torch.uint8, device="cuda")
            keep.append(t); outs.append(t); ostr.append(stride)
        dp.append(outs); ds.append(ostr)
    plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        wy, wu, wv = synth.split(want[i], ow, oh)
        for k, wpl in enumerate((wy, wu, wv)):
            got = dp[i][k].cpu().numpy()
            bad = got[:, : wpl.shape[1]] != wpl
            if bad.any():
                ys, xs = np.nonzero(bad)
                ends = sorted((sp[f][q] + ss[f][q] * ((ih if q == 0 else (ih + 1) // 2) - 1) + (iw if q == 0 else (iw + 1) // 2), f, q)
                              for f in range(n) for q in range(3))
                raise AssertionError("frame %d plane %d: %d bad, rows %d..%d cols %d..%d; src planes %s; highest end: frame %d plane %d" % (
                    i, k, bad.sum(), ys.min(), ys.max(), xs.min(), xs.max(), [[hex(p) for p in f] for f in sp], ends[-1][1], ends[-1][2]))
            assert (got[:, wpl.shape[1]:] == 7).all(), "wrote into the stride padding"


# ---- BASELINE.json full sizes: size-independent properties ---------------------------------

def test_full_size_constant_planes_stay_constant(tts):
    import torch
    plan = tts.Plan.from_vf(PORTRAIT, 1920, 1080)
    for val in (0, 16, 128, 235, 255):
        src = torch.full((4, plan.in_bytes), val, dtype=torch.uint8, device="cuda")
        out = plan.run_batch(src)
        assert int(out.min()) == val and int(out.max()) == val


def test_full_size_crop_equals_uncropped_then_sliced(tts):
    """scale+crop must equal the crop of the full virtual scale (fusing the crop is invisible)."""
    import torch
    src = torch.from_numpy(synth.frames("random", 1920, 1080, 2)).cuda()
    fused = tts.Plan.from_vf(PORTRAIT, 1920, 1080).run_batch(src).cpu().numpy()
    full = tts.Plan.from_vf("scale=3413:1920", 1920, 1080).run_batch(src).cpu().numpy()
    for i in range(2):
        fy, fu, fv = synth.split(full[i], 3413, 1920)
        cy, cu, cv = synth.split(fused[i], 1080, 1920)
        assert np.array_equal(cy, fy[:, 1166:1166 + 1080])
        assert np.array_equal(cu, fu[:, 583:583 + 540])
        assert np.array_equal(cv, fv[:, 583:583 + 540])


def test_full_size_frames_are_independent(tts):
    """Every frame of a 300-frame batch equals the same frame run alone (checksum of checksums)."""
    import torch
    plan = tts.Plan.from_vf(PORTRAIT, 1920, 1080)
    base = torch.from_numpy(synth.frames("random", 1920, 1080, 3)).cuda()
    single = plan.run_batch(base)
    batch = base.repeat(100, 1)
    out = plan.run_batch(batch)
    torch.cuda.synchronize()
    assert torch.equal(out.view(100, 3, -1), single.unsqueeze(0).expand(100, -1, -1))


def test_full_size_thumbnails_match_oracle_sample(tts, oracle):
    import torch
    plan = tts.Plan.from_vf("scale=320:568:flags=area", 1080, 1920)
    src = synth.frames("random", 1080, 1920, 2)
    got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.run_chain(plan.geom, src))


# ---- committed golden vectors and the multi-clip path ---------------------------------------------

import glob as _glob
import os as _os

_GOLDEN = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(p for p in _glob.glob(_os.path.join(_GOLDEN, "*.npz")) if "filter_tables" not in p))
def test_hip_matches_golden_vectors(tts, path):
    import torch
    z = np.load(path)
    g = dict(zip([str(k) for k in z["geom_keys"]], [int(v) for v in z["geom"]]))
    geom = tts.Geom(**g)
    plan = tts.Plan(geom)
    for key in z.files:
        if key == "out_rgbseed":                          # packed rgb24 in: seeded random bytes (tests/golden/make_golden.py)
            src = np.random.default_rng(synth.SEED).integers(0, 256, (1, g["in_w"] * g["in_h"] * 3), dtype=np.uint8)
        elif key.startswith("out_"):
            src = synth.frames(key[4:], g["in_w"], g["in_h"], 1)
        else:
            continue
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()[0]
        assert np.array_equal(got, z[key]), "%s %s" % (path, key)


def test_plan_tables_match_oracle(tts, oracle):
    plan = tts.Plan.from_vf(PORTRAIT, 1920, 1080)
    for cls, (sw, dw, sh, dh) in enumerate([(1920, 3413, 1080, 1920), (960, 1707, 540, 960)]):
        for axis, (s, d, one) in enumerate([(sw, dw, 1 << 14), (sh, dh, 1 << 12)]):
            c, p = plan.filter(cls, axis)
            oc, op = oracle.filter_table(s, d, one, 0)
            assert np.array_equal(c, oc) and np.array_equal(p, op)


def test_clip_shard_runner(tts, oracle):
    """One rank's share of a small mixed corpus through tiktoken_amd.shard.run_shard."""
    import torch
    from tiktoken_amd.shard import Clip, assign, run_shard
    clips = [Clip("a", 128, 72, 5), Clip("b", 192, 108, 3), Clip("c", 128, 72, 9)]
    vf = "scale=108:192:force_original_aspect_ratio=increase,crop=108:192"
    shards = assign(clips, 2, (108, 192))

    def frames(c, first, n):
        return torch.from_numpy(synth.frames("random", c.width, c.height, c.frames)[first:first + n]).cuda()

    sums = {}
    for r in range(2):
        sums.update(run_shard(clips, shards[r], vf, frames, batch=4))
    for c in clips:
        g = tts.parse_vf(vf, c.width, c.height)
        want = int(oracle.run_chain(g, synth.frames("random", c.width, c.height, c.frames)).astype(np.int64).sum())
        assert sums[c.name] == want


# The LDS-staged kernel's own corner cases: (chain, iw, ih, frames, expected kernel kind of the luma class).
# kind: 1 = register strip walk, 2 / 3 = LDS-staged with one / two DMAs per source row (include/ttscale.h).
LDS_CASES = [
    ("scale=136:96", 100, 60, 70, 2),                      # 17 strips per row: a wave wraps over 4-5 plane rows (pieces)
    ("scale=24:40", 18, 30, 300, 1),                       # 3 strips per luma row: 22 pieces per wave -> register walk
    ("scale=72:40", 54, 30, 100, 2),                       # 9 strips per luma row: 8 pieces per wave, the most the LDS walk takes
    ("scale=16:16", 12, 12, 500, None),                    # 2 luma strips, 1 chroma strip per row: too many pieces -> register walk
    ("scale=320:568:flags=area", 1080, 1920, 3, 3),        # 5-tap per-column windows, 40 strips per row, 1.1 KB runs: two DMAs
    ("scale=256:144:flags=area", 960, 540, 9, 3),          # 3.75x area: source run of a wave needs two DMAs
    ("scale=200:112:flags=area", 1400, 784, 5, 3),         # 7x area: a wave's runs take four DMAs per row
    ("scale=320:180:flags=area", 1280, 720, 4, 3),         # 4x area, 4 taps: four DMAs
    ("scale=3413:1920,crop=1080:1920", 3840, 2160, 1, 2),  # 6-tap bicubic downscale, 8-byte windows
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1024:1920", 1920, 1080, 2, 2),   # aligned rows: unpaced
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1920, 1080, 9, 2),   # paced, 1024-thread workgroups
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,pad=1200:2000:60:40", 1280, 720, 2, 2),  # paced into a padded canvas
]


# Grouped 12-byte windows (5-6 taps, ratios up to 4/3): (chain, iw, ih, frames, expected luma kernel)
GROUPED6_CASES = [
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 3840, 2160, 2, "strip_lds<6,6,grouped,1,1>"),   # BASELINE config 4
    ("scale=3413:1920,crop=1080:1920", 3840, 2160, 1, "strip_lds<6,6,grouped,1,1>"),
    ("scale=640:ih", 720, 96, 5, "strip_lds<6,4,grouped,1,1>"),       # 1.125x horizontally only: ring of 4
    ("scale=640:320", 720, 400, 3, "strip_lds<6,6,grouped,1,1>"),     # whole width: the last group sits against the row end
    ("scale=600:300", 720, 400, 3, "strip_lds<6,6,grouped,1,1>"),     # 1.2x
    ("scale=540:300", 720, 400, 3, "strip_lds<6,6,grouped,1,1>"),     # exactly 4/3: the fourth column's first tap is byte 4
    ("scale=640:250", 720, 400, 3, "strip_lds<6,8,grouped,1,1>"),     # 1.6x vertically: ring of 8
    ("scale=520:300", 720, 400, 3, "strip_lds<6,6,grouped2,1,1>"),    # 1.38x: four columns' windows pass 12 bytes, two fit
    ("scale=360:200", 720, 400, 3, None),                             # 2x: 8 taps, two columns a window
    ("scale=214:120:flags=area", 720, 400, 3, None),                  # 3.4x area (BASELINE configs[2]'s ratio): 5 taps, two columns a window
    ("scale=320:568:flags=area", 1080, 1920, 2, "strip_lds<6,6,grouped2,2,1>"),   # BASELINE configs[2]
    ("scale=1920:1080", 3840, 2160, 1, "strip_lds<8,8,grouped2,2,4>"),            # 4K -> 1080p: four source rows in flight
    ("scale=960:400", 1920, 400, 2, None),                            # 2x horizontally only: ring of 4
]


@pytest.mark.parametrize("vf,iw,ih,n,name", GROUPED6_CASES)
def test_grouped_12_byte_windows(tts, oracle, vf, iw, ih, n, name):
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih)
    if name is None:
        assert "grouped2" in plan.kernel_name(0), plan.kernel_name(0)
    else:
        assert plan.kernel_name(0) == name
    for pat in ("random", "impulse"):
        src = synth.frames(pat, iw, ih, n)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        assert np.array_equal(got, oracle.run_chain(plan.geom, src)), pat


def test_grouped_12_byte_windows_pointer_api(tts, oracle):
    """Whole-width 1.125x downscale through tts_run, every plane its own tight allocation: the last group's
    window hangs over the row's end, and the staged runs must still end with the plane."""
    import torch
    iw, ih, n = 720, 400, 5
    plan = tts.Plan.from_vf("scale=640:320", iw, ih)
    assert "grouped" in plan.kernel_name(0)
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    keep, sp, ss, dp, ds = [], [], [], [], []
    for i in range(n):
        planes = [torch.from_numpy(np.ascontiguousarray(pl)).cuda() for pl in synth.split(src[i], iw, ih)]
        outs = [torch.full((oh if k == 0 else (oh + 1) // 2, ow if k == 0 else (ow + 1) // 2), 7, dtype=torch.uint8, device="cuda") for k in range(3)]
        keep += planes + outs
        sp.append([t.data_ptr() for t in planes]); ss.append([t.shape[1] for t in planes])
        dp.append(outs); ds.append([t.shape[1] for t in outs])
    plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        for k, w in enumerate(synth.split(want[i], ow, oh)):
            assert np.array_equal(dp[i][k].cpu().numpy(), w), (i, k)


@pytest.mark.parametrize("vf,iw,ih,n,kind", LDS_CASES)
def test_lds_kernel_variants(tts, oracle, vf, iw, ih, n, kind):
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih)
    if kind is not None:
        assert plan.kernel_kind(0) == kind, "luma class runs kernel kind %d, expected %d" % (plan.kernel_kind(0), kind)
    src = synth.frames("random", iw, ih, n)
    got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    want = oracle.run_chain(plan.geom, src)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("knob,val", [("TTS_PACED", "1"), ("TTS_PACED", "0"), ("TTS_LDS_BLOCK", "64"), ("TTS_LDS_BLOCK", "704"),
                                      ("TTS_STRIP_D", "2"), ("TTS_NO_LDS", "1"), ("TTS_NO_G4", "1"), ("TTS_NO_G6", "1"), ("TTS_NO_MERGE", "1"), ("TTS_COL_NC", "1"), ("TTS_WAVE_BASE", "1"), ("TTS_NO_DMA4", "1"),
                                      ("TTS_COL_FILL", "1"), ("TTS_COL_FILL", "80"), ("TTS_COL_NC_LAUNCH", "1"), ("TTS_COL_NC_LAUNCH", "2"),
                                      ("TTS_COL_D4", "0"), ("TTS_COL_D4", "2"), ("TTS_COL_MODEL", "0"), ("TTS_COL_MODEL", "2")])
def test_tuning_knobs_do_not_change_pixels(tts, oracle, monkeypatch, knob, val):
    import torch
    monkeypatch.setenv(knob, val)
    for vf, iw, ih in (("scale=355:267", 200, 150), ("scale=120:90:flags=area", 200, 150), ("scale=136:96", 100, 60),
                       ("scale=176:132", 200, 150), ("scale=67:49", 200, 150), ("scale=40:30", 200, 150)):
        plan = tts.Plan.from_vf(vf, iw, ih)
        src = synth.frames("random", iw, ih, 21)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        assert np.array_equal(got, oracle.run_chain(plan.geom, src)), (knob, val, vf)


D4_CASES = [
    # the reference's thumbnail chain (src/services/video.js:329-332) on landscape, portrait and 4K sources
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1920, 1080, 2),
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1080, 1920, 2),
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 3840, 2160, 1),
    ("scale=1280:720", 3840, 2160, 1),                       # 13 taps, two columns per lane
    ("scale=213:121", 1283, 721, 5),                         # odd everything: frames start at every byte phase, last strip slid
    ("crop=1001:701:13:7,scale=167:117", 1283, 721, 3),      # odd pre-crop origin
    ("scale=67:49", 200, 150, 21), ("scale=40:30", 200, 150, 21), ("scale=33:201", 333, 801, 4),
    ("scale=320:180:flags=area", 3840, 2160, 1),             # 12x with 13 taps: adjacent windows lie 12 bytes apart, no shared span
]


@pytest.mark.parametrize("vf,iw,ih,n", D4_CASES)
@pytest.mark.parametrize("nc_launch", ["0", "1", "2"])
def test_column_walk_dot4_form(tts, oracle, monkeypatch, vf, iw, ih, n, nc_launch):
    """column_lds_kernel's v_dot4 form (two signed-byte digits per tap, pixels biased by 128, byte phase folded
    into the digit tables; TTS_COL_D4) against the oracle, bit for bit, at one and two columns per lane."""
    import torch
    monkeypatch.setenv("TTS_COL_D4", "2")                     # wherever the digit tables exist (the product: above 8 tap pairs)
    monkeypatch.setenv("TTS_COL_NC_LAUNCH", nc_launch)
    plan = tts.Plan.from_vf(vf, iw, ih)
    if plan.kernel_kind(0) in (4, 5) and "area" not in vf:
        assert "dot4" in plan.kernel_name(0), plan.kernel_name(0)
    src = synth.frames("random", iw, ih, n)
    got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.run_chain(plan.geom, src))
    src = synth.frames("impulse", iw, ih, n)                 # extreme pixels: 0 / 255 against the largest taps
    got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.run_chain(plan.geom, src))


def test_pointer_api_shuffled_placements(tts, oracle):
    """tts_run again, many times, with the planes (each its own allocation) landing at different
    addresses and in different orders: results must not depend on where a plane lies, what lies
    behind it, or which plane ends the launch's span."""
    import random
    import torch
    iw, ih, n = 1283, 721, 3
    plan = tts.Plan.from_vf("scale=640:360:flags=area", iw, ih)
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    rng = random.Random(7)
    for rep in range(25):
        keep = [torch.full((rng.randrange(1, 3000) * 512,), 0xAB, dtype=torch.uint8, device="cuda") for _ in range(rng.randrange(0, 6))]
        slots = {}
        order = list(range(n))
        rng.shuffle(order)
        for i in order:
            planes, strides, outs, ostr = [], [], [], []
            for pl in synth.split(src[i], iw, ih):
                t = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
                keep.append(t); planes.append(t.data_ptr()); strides.append(pl.shape[1])
                if rng.random() < 0.3:
                    keep.append(torch.full((rng.randrange(1, 64) * 512,), 0xCD, dtype=torch.uint8, device="cuda"))
            for k in range(3):
                w_, h_ = (ow, oh) if k == 0 else ((ow + 1) // 2, (oh + 1) // 2)
                t = torch.full((h_, w_), 7, dtype=torch.uint8, device="cuda")
                keep.append(t); outs.append(t); ostr.append(w_)
            slots[i] = (planes, strides, outs, ostr)
        sp, ss, dp, ds = ([slots[i][j] for i in range(n)] for j in range(4))
        plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
        torch.cuda.synchronize()
        for i in range(n):
            for k, w in enumerate(synth.split(want[i], ow, oh)):
                assert np.array_equal(dp[i][k].cpu().numpy(), w), "rep %d frame %d plane %d" % (rep, i, k)


@pytest.mark.parametrize("vf,iw,ih,n", [
    (PORTRAIT, 1920, 1080, 1400),                  # the paced 1024-thread workgroups, 4.35 GB in and out
    (THUMB_REF, 1920, 1080, 1400),                 # the column walk
    (PORTRAIT, 3840, 2160, 350),                   # grouped 12-byte windows
])
def test_batch_longer_than_32_bit_offsets_other_kernels(tts, oracle, vf, iw, ih, n):
    """The same for the other walk kernels: every wave addresses from its own first frame."""
    import torch
    distinct = 4
    plan = tts.Plan.from_vf(vf, iw, ih)
    assert n * plan.in_bytes > 0xFF000000
    base = synth.frames("random", iw, ih, distinct)
    want = torch.from_numpy(oracle.run_chain(plan.geom, base)).cuda()
    src = torch.from_numpy(base).cuda().repeat((n + distinct - 1) // distinct, 1)[:n].contiguous()
    out = plan.run_batch(src)
    torch.cuda.synchronize()
    idx = torch.arange(n, device="cuda") % distinct
    assert bool((out == want[idx]).all())
    del src, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("chunks", [0, 1])
def test_batch_longer_than_32_bit_offsets(tts, oracle, monkeypatch, chunks):
    """1,500 frames of 1080x1920 (4.7 GB of source) pass what 32-bit lane offsets reach from one base:
    every wave addresses from its own first frame (one launch), or -- TTS_SPAN_CHUNKS=1, the path kept for
    huge frame pitches -- the batch is cut into launches; every frame must come out right either way."""
    import torch
    monkeypatch.setenv("TTS_SPAN_CHUNKS", str(chunks))
    iw, ih, n, distinct = 1080, 1920, 1500, 6
    plan = tts.Plan.from_vf("scale=320:568:flags=area", iw, ih)
    assert n * plan.in_bytes > 0xFF000000
    base = synth.frames("random", iw, ih, distinct)
    want = torch.from_numpy(oracle.run_chain(plan.geom, base)).cuda()
    src = torch.from_numpy(base).cuda().repeat((n + distinct - 1) // distinct, 1)[:n].contiguous()
    out = plan.run_batch(src)
    torch.cuda.synchronize()
    idx = torch.arange(n, device="cuda") % distinct
    assert bool((out == want[idx]).all())


def test_more_planes_than_grid_z(tts, oracle):
    """40,000 tiny frames with a pad and a range step: the fill / range kernels index planes by grid.z
    (65,535 at most), so the launch is repeated over frame ranges."""
    import torch
    iw, ih, n, distinct = 16, 16, 40000, 5
    plan = tts.Plan.from_vf("scale=24:20,pad=32:24:-1:-1:color=white,format=yuv420p,format=yuvj420p", iw, ih)
    base = synth.frames("random", iw, ih, distinct)
    want = torch.from_numpy(oracle.run_chain(plan.geom, base)).cuda()
    src = torch.from_numpy(base).cuda().repeat(n // distinct, 1).contiguous()
    out = plan.run_batch(src)
    assert bool((out == want[torch.arange(n, device="cuda") % distinct]).all())
    # the generic kernel likewise
    plan = tts.Plan.from_vf("scale=2:2", iw, ih)
    assert plan.kernel_name(0) == "generic"
    want = torch.from_numpy(oracle.run_chain(plan.geom, base)).cuda()
    out = plan.run_batch(src)
    assert bool((out == want[torch.arange(n, device="cuda") % distinct]).all())


# The column walk (9..64 taps per axis): (chain, iw, ih, frames, expected kernel kind of the luma class); 4 / 5 = one /
# two LDS-DMAs per source row, 0 = generic kernel (more than 64 taps).
COLUMN_CASES = [
    (THUMB_REF, 1920, 1080, 9, 4),                         # src/services/video.js:329-332 on 1080p: 6x, 25 taps (12 tap pairs after trimming)
    (THUMB_REF, 3840, 2160, 2, 4),                         # ... on 4K: 12x, 49 taps (26 tap pairs, ring of 52 rows)
    (THUMB_REF, 720, 1280, 70, 4),                         # ... on a portrait clip: 101 columns, waves wrap over planes
    (THUMB_REF + ",format=yuvj420p", 1080, 1080, 3, 4),    # ... square source, then the mjpeg range pass
    ("scale=160:90", 1920, 1080, 5, 4),
    ("scale=40:ih", 640, 48, 33, 5),                       # 16x horizontally only: runs of 1.1 KB, vertical taps = 1
    ("scale=iw:30", 64, 480, 33, 4),                       # 16x vertically only
    ("scale=320:180:flags=area", 3840, 2160, 2, 5),        # 12x area: 13 taps, two columns per lane: 1.5 KB runs
    ("scale=20:12", 200, 120, 200, 4),                     # 20-column planes: 3-4 pieces per wave
    ("crop=1280:720:320:180,scale=160:90", 1920, 1080, 3, 4),   # pre-crop: windows start inside the rows
    ("scale=720:1280", 2160, 3840, 1, 4),                  # web compress of a 4K portrait clip: 3x, 13 taps (8 tap pairs, ring of 16)
    ("scale=480:270", 1920, 1080, 4, 4),                   # 4x: 17 taps -> 16 tap pairs
    ("scale=64:36", 1920, 1080, 2, 0),                     # 30x: 121 taps -> generic kernel
    # two columns per lane (8 tap pairs): odd widths slide the last strip left; 3 / 2-column planes fall back to one
    ("scale=213:121", 640, 360, 7, 4),
    ("scale=3:50", 10, 150, 40, None),
    ("scale=1280:720", 3840, 2160, 1, 4),
]


@pytest.mark.parametrize("vf,iw,ih,n,kind", COLUMN_CASES)
def test_column_kernel_variants(tts, oracle, vf, iw, ih, n, kind):
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih)
    if kind is not None:
        assert plan.kernel_kind(0) == kind, "luma class runs kernel kind %d, expected %d" % (plan.kernel_kind(0), kind)
    if vf in ("scale=213:121", "scale=1280:720", "scale=720:1280"):
        assert plan.kernel_name(0) == "column_lds<6,12,1,2,4>"      # 11-12 taps after trimming: 6 tap pairs, two columns per lane
    if vf == THUMB_REF and (iw, ih) == (1920, 1080):
        assert plan.kernel_name(0) == "column_lds<12,24,1,2,4,dot4>"   # 23 taps after trimming; two columns per lane up to 12 tap pairs; v_dot4 form above 8
    for pat in ("random", "impulse"):
        src = synth.frames(pat, iw, ih, n)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        assert np.array_equal(got, oracle.run_chain(plan.geom, src)), pat


def test_plan_shared_by_host_threads(tts, oracle):
    """SURVEY.md 8b threading: one immutable plan, several host threads, each with its own HIP stream
    and its own frames, all in flight together."""
    import threading
    import torch
    iw, ih, n, workers = 320, 180, 12, 4
    plan = tts.Plan.from_vf("scale=568:320", iw, ih)
    srcs = [synth.frames("random", iw, ih, n) + np.uint8(w) for w in range(workers)]
    outs, errs = [None] * workers, []

    def work(w):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                d_src = torch.from_numpy(srcs[w]).cuda()
                res = None
                for _ in range(20):                       # keep the streams busy at the same time
                    res = plan.run_batch(d_src, stream=st.cuda_stream)
                st.synchronize()
                outs[w] = res.cpu().numpy()
        except Exception as e:                            # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for w in range(workers):
        assert np.array_equal(outs[w], oracle.run_chain(plan.geom, srcs[w])), w


def test_runs_under_hip_graph_capture(tts, oracle):
    """tts_run_batch allocates nothing and only launches kernels, so a serving loop can capture it once
    and replay it: capture on a side stream, replay with new frames in the same buffers."""
    import torch
    iw, ih, n = 320, 180, 6
    plan = tts.Plan.from_vf("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p,format=yuvj420p", 180, 320)
    plan2 = tts.Plan.from_vf("scale=568:320", iw, ih)
    a = synth.frames("random", 180, 320, n)
    b = synth.frames("random", iw, ih, n)
    d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    o_a = torch.zeros((n, plan.out_bytes), dtype=torch.uint8, device="cuda")
    o_b = torch.zeros((n, plan2.out_bytes), dtype=torch.uint8, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = torch.cuda.current_stream().cuda_stream
        plan.run_batch(d_a, o_a, stream=st)              # pad border + column walk + range step: ONE launch since round 2
        plan2.run_batch(d_b, o_b, stream=st)
    for rep in range(3):
        a2 = np.roll(a, rep + 1, axis=0) ^ np.uint8(rep)
        b2 = np.roll(b, rep + 1, axis=0) ^ np.uint8(3 * rep)
        d_a.copy_(torch.from_numpy(a2)); d_b.copy_(torch.from_numpy(b2))
        o_a.zero_(); o_b.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(o_a.cpu().numpy(), oracle.run_chain(plan.geom, a2)), rep
        assert np.array_equal(o_b.cpu().numpy(), oracle.run_chain(plan2.geom, b2)), rep


def test_run_host_pipeline_of_many_chunks(tts, oracle, monkeypatch):
    """tts_run_host moves a clip in chunks over two slots / two streams: with 1 MB chunks 37 small frames
    take seven of them, the last one short; a preallocated output is filled in place."""
    monkeypatch.setenv("TTS_HOST_CHUNK_MB", "1")
    iw, ih, n = 320, 180, 37
    plan = tts.Plan.from_vf("scale=426:240", iw, ih)
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    assert np.array_equal(plan.run_host(src), want)
    out = np.zeros((n, plan.out_bytes), np.uint8)
    assert plan.run_host(src, out) is out and np.array_equal(out, want)
    with pytest.raises(tts.TtsError):
        plan.run_host(src, np.zeros(5, np.uint8))


@pytest.mark.parametrize("vf,iw,ih,ss,ds", [
    (PORTRAIT, 1920, 1080, (1984, 992), (1088, 576)),          # decoder-style line sizes in, 64-byte-aligned rows out
    ("scale=w=720:h=1280", 480, 854, (480, 240), (768, 384)),
    (THUMB_REF, 1283, 721, (1344, 704), (320, 160)),           # odd source, padded in, packed out (column walk + pad)
    ("scale=320:568:flags=area", 1080, 1920, (1088, 544), (384, 192)),
])
def test_strided_batch_api(tts, oracle, vf, iw, ih, ss, ds):
    """tts_run_batch_strided: packed batches whose rows carry padding (AVFrame.linesize style)."""
    import torch
    n = 3
    plan = tts.Plan.from_vf(vf, iw, ih)
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    sp, dp = plan.strided_bytes("in", *ss), plan.strided_bytes("out", *ds)

    def planes(buf, w, h, st):                         # views of the three planes of every frame of a strided buffer
        ch, cw = (h + 1) // 2, (w + 1) // 2
        y = buf[:, : st[0] * h].reshape(-1, h, st[0])
        u = buf[:, st[0] * h: st[0] * h + st[1] * ch].reshape(-1, ch, st[1])
        v = buf[:, st[0] * h + st[1] * ch: st[0] * h + 2 * st[1] * ch].reshape(-1, ch, st[1])
        return (y, w), (u, cw), (v, cw)

    s_host = np.full((n, sp), 0x5A, np.uint8)
    for i in range(n):
        for (view, w), pl in zip(planes(s_host, iw, ih, ss), synth.split(src[i], iw, ih)):
            view[i, :, :w] = pl
    d_dev = torch.full((n, dp), 7, dtype=torch.uint8, device="cuda")
    plan.run_batch_strided(torch.from_numpy(s_host).cuda(), ss, d_dev, ds, n)
    got = d_dev.cpu().numpy()
    for i in range(n):
        for (view, w), pl in zip(planes(got, ow, oh, ds), synth.split(want[i], ow, oh)):
            assert np.array_equal(view[i, :, :w], pl), (vf, i)
            assert (view[i, :, w:] == 7).all(), "wrote into the row padding"
    with pytest.raises(tts.TtsError):
        plan.run_batch_strided(torch.from_numpy(s_host).cuda(), (iw - 1, ss[1]), d_dev, ds, n)


def test_plan_refuses_another_device(tts):
    """A plan's tables live on the device it was made on; the run calls refuse any other current device
    instead of reading across the fabric (needs two GPUs: skipped on a one-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU")
    plan = tts.Plan.from_vf("scale=160:90", 320, 180)
    with torch.cuda.device(1):
        src = torch.zeros((1, plan.in_bytes), dtype=torch.uint8, device="cuda:1")
        dst = torch.zeros((1, plan.out_bytes), dtype=torch.uint8, device="cuda:1")
        with pytest.raises(tts.TtsError, match="INVALID_INPUT"):
            plan.run_batch(src, dst)
        torch.cuda.synchronize()


def test_host_frames_through_tts_run(tts, oracle):
    """tts_frame.mem = TTS_MEM_HOST: tts_run stages host frames (padded rows, scattered planes) itself."""
    iw, ih, n = 160, 90, 7
    plan = tts.Plan.from_vf("scale=284:160", iw, ih)
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    keep, sp, dp, ss, ds = [], [], [], [], []
    for i in range(n):
        planes = []
        for k, pl in enumerate(synth.split(src[i], iw, ih)):
            buf = np.full((pl.shape[0], pl.shape[1] + 5 + k), 0xEE, np.uint8)     # padded rows
            buf[:, : pl.shape[1]] = pl
            planes.append(buf)
        outs = [np.full((oh, ow + 3), 0xA5, np.uint8), np.full(((oh + 1) // 2, (ow + 1) // 2 + 1), 0xA5, np.uint8),
                np.full(((oh + 1) // 2, (ow + 1) // 2 + 1), 0xA5, np.uint8)]
        keep.append((planes, outs))
        sp.append([b.ctypes.data for b in planes]); ss.append([b.shape[1] for b in planes])
        dp.append([b.ctypes.data for b in outs]); ds.append([b.shape[1] for b in outs])
    plan.run_frames(sp, dp, ss, ds, host=True)
    for i in range(n):
        outs = keep[i][1]
        got = np.concatenate([outs[0][:, :ow].ravel(), outs[1][:, : (ow + 1) // 2].ravel(), outs[2][:, : (ow + 1) // 2].ravel()])
        assert np.array_equal(got, want[i])
        assert (outs[0][:, ow:] == 0xA5).all() and (outs[1][:, (ow + 1) // 2:] == 0xA5).all()
    # mixing host and device frames in one call is an error, not a crash
    import torch
    d = torch.zeros(plan.out_bytes, dtype=torch.uint8, device="cuda")
    S, D = (tts.Frame * 1)(), (tts.Frame * 1)()
    for k in range(3):
        S[0].plane[k], S[0].stride[k] = sp[0][k], ss[0][k]
        D[0].plane[k], D[0].stride[k] = d.data_ptr(), ow
    S[0].width, S[0].height, D[0].width, D[0].height = iw, ih, ow, oh
    S[0].mem, D[0].mem = 1, 0
    assert tts._lib.lib().tts_run(plan._h, S, D, 1, None) == -1


@pytest.mark.parametrize("vf,iw,ih,family", [
    ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuvj420p", 1920, 1080, "column_lds"),   # many taps
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=yuvj420p", 1280, 720, "strip_lds<4,4,grouped"),       # four taps: the hot kernel's RNG form
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=yuvj420p", 3840, 2160, "strip_lds<6,6,grouped"),      # 1.125x shrink: 12-byte windows, four columns each
    ("scale=720:1280,format=yuvj420p", 1080, 1920, "strip_lds<6,6,grouped,"),                                                          # the web box, 1.5x
    ("scale=1920:1080,format=yuvj420p", 3840, 2160, "strip_lds<8,8,grouped2"),                                                         # 2x: 8 taps, four rows in flight
    ("scale=720:1280,pad=800:1300:-1:-1:color=red,format=yuvj420p", 480, 854, "strip_lds<4,4,grouped"),                                # R1 + a pad on the yuvj420p link
    ("scale=640:360:flags=area,format=yuvj420p", 1280, 720, "column_lds"),                                                           # per-column windows have no RNG form: the column walk
    ("scale=320:568:flags=area,format=yuvj420p", 1080, 1920, "strip_lds<6,6,grouped2"),                                                # 3.4x area thumbnails: two columns a window
    ("scale=500:300:flags=bilinear,format=yuvj420p", 1280, 720, ""),                                                                   # whatever the planner picks
    ("scale=12:10,format=yuvj420p", 8, 8, "generic"),                                                                                  # too small for the walks
])
def test_range_step_inside_the_scale(tts, oracle, vf, iw, ih, family):
    """`scale=...,format=yuvj420p` with nothing pinning yuv420p in between: swscale converts the range on the scale's 15-bit rows
    (oracle: tto_scale_plane_range).  The grouped-window strip kernels and the column walk carry the step (",range15" in their
    names), the general kernel behind them; the pinned form of the same chain gives other bytes."""
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih)
    assert plan.geom.sc_range == 1 and plan.geom.out_range == 0
    name = plan.kernel_name(0)
    assert name.startswith(family) and (family == "generic" or name.endswith(",range15>")), name
    pinned = tts.Plan.from_vf(vf.replace(",format=yuvj420p", ",format=yuv420p,format=yuvj420p"), iw, ih)
    assert pinned.geom.sc_range == 0 and pinned.geom.out_range == 1 and "range15" not in pinned.kernel_name(0)
    for pat in ("random", "impulse", "limited"):
        src = synth.frames(pat, iw, ih, 2 if iw > 2000 else 3)
        d = torch.from_numpy(src).cuda()
        got = plan.run_batch(d).cpu().numpy()
        want = oracle.run_chain(plan.geom, src)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d %s [%s]: %d bytes differ, first at %d" % (vf, iw, ih, pat, name, bad.size, bad[0])
        other = pinned.run_batch(d).cpu().numpy()
        assert np.array_equal(other, oracle.run_chain(pinned.geom, src))
        if pat == "random" and iw > 16:   # (a tiny picture may well agree)
            assert not np.array_equal(got, other)


THUMB_PINNED = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p"
FULL_RANGE_SOURCE_CASES = [
    ("scale=w=720:h=1280", 480, 854, (0, 0)),                                    # R1: no format anywhere, the link stays full: no step at all
    (THUMB_PINNED, 720, 1280, (2, 0)),                                            # R2: the scale writes the pinned yuv420p: full -> limited inside its context
    (THUMB_PINNED + ",format=yuvj420p", 720, 1280, (2, 1)),                       # ... and the mjpeg encoder's pass back to full range at the end: both steps
    (THUMB_PINNED + ",format=yuvj420p", 1920, 1080, (2, 1)),
    ("scale=320:180,pad=400:200:-1:-1:color=red", 1280, 720, (0, 0)),             # the pad on a full-range link: its colour through RGB_TO_Y_JPEG & co.
    ("crop=640:360:100:50,scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=yuv420p", 1280, 720, (2, 0)),   # four taps
    ("format=yuv420p", 64, 48, (0, 2)),                                           # no scale: a pass of its own, full -> limited
    ("crop=32:20:4:2,pad=64:48:-1:-1:color=white,format=yuv420p", 64, 48, (0, 2)),
]


@pytest.mark.parametrize("vf,iw,ih,steps", FULL_RANGE_SOURCE_CASES)
def test_full_range_source(tts, oracle, vf, iw, ih, steps):
    """`-pix_fmt yuvj420p -i ...` (what some phone cameras record): the same planes, so the same kernels -- what changes is
    where range steps sit.  The reference's thumbnail chain pins format=yuv420p: its scale then converts full -> limited inside
    its context (tts_geom.sc_range), and the mjpeg encoder's yuvj420p is a second step at the end (out_range)."""
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt="yuvj420p")
    assert (plan.geom.sc_range, plan.geom.out_range) == steps and plan.geom.in_fmt == 0
    limited = tts.Plan.from_vf(vf, iw, ih)
    if steps[0]:
        assert "range15" in plan.kernel_name(0) or plan.kernel_name(0) == "generic", plan.kernel_name(0)
    for pat in ("random", "impulse", "zone"):
        src = synth.frames(pat, iw, ih, 3)
        d = torch.from_numpy(src).cuda()
        got = plan.run_batch(d).cpu().numpy()
        want = oracle.run_chain(plan.geom, src)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d %s [%s]: %d bytes differ, first at %d" % (vf, iw, ih, pat, plan.kernel_name(0), bad.size, bad[0])
        same = np.array_equal(got, limited.run_batch(d).cpu().numpy())
        assert same == (steps == (0, 0) and "pad" not in vf), "a limited-range source of the same bytes: %s" % same


@pytest.mark.parametrize("no_union", [0, 1])
@pytest.mark.parametrize("vf,iw,ih,n", [
    ("scale=640:360", 1280, 720, 3),                      # exact 2x: a lane's four 12-byte windows one dword apart
    ("scale=1920:1080", 3840, 2160, 2),                   # ... the four-rows-in-flight instantiation
    ("scale=960:540,crop=900:500:30:20", 1920, 1080, 2),  # a crop behind it: strips that start off the lattice's origin
    ("scale=642:360", 1280, 720, 2),                      # not quite 2x: the lattice holds for some waves only
])
def test_exact_2x_shrink_one_run_window_read(tts, oracle, monkeypatch, vf, iw, ih, n, no_union):
    """strip_lds_kernel, two columns a 12-byte window: waves whose lanes all have their four windows one dword apart read the
    run of 8 dwords once (the LDS bank conflicts of the exact 2x shrink); TTS_NO_UNION=1 keeps the window-by-window
    read.  Both against the oracle."""
    import torch
    monkeypatch.setenv("TTS_NO_UNION", str(no_union))
    plan = tts.Plan.from_vf(vf, iw, ih)
    assert plan.kernel_name(0).startswith("strip_lds<8,8,grouped2"), plan.kernel_name(0)
    for pat in ("random", "impulse"):
        src = synth.frames(pat, iw, ih, n)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        want = oracle.run_chain(plan.geom, src)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d %s: %d bytes differ, first at %d" % (vf, iw, ih, pat, bad.size, bad[0])
