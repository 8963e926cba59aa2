"""GPU parity of the packed-RGB output: libttscale through the C ABI vs the CPU oracle's
restatement of swscale's yuv2rgb C path, bit-exact (0 LSB).  The oracle keeps swscale's table construction
(libswscale/yuv2rgb.c), the kernels evaluate the closed form of the same tables: two formulations that must
agree byte for byte.  PARITY UNPINNED against real FFmpeg like the rest (oracle/swscale_oracle.c)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

PORTRAIT = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
THUMB = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black"

CASES = [
    (THUMB + ",format=rgb24", 720, 1280),              # src/services/video.js:329-332 handing RGB to the next stage
    (THUMB + ",format=rgb24", 1280, 720),
    (THUMB + ",format=yuv420p,format=rgb24", 720, 1280),   # the reference's literal chain, then a conversion pass
    (THUMB + ",format=bgra", 1080, 1080),
    (PORTRAIT + ",format=rgb24", 1920, 1080),          # BASELINE configs[1] + the north_star's colour conversion
    (PORTRAIT + ",format=rgb24", 1280, 720),
    (PORTRAIT + ",format=rgba", 3840, 2160),
    ("scale=w=720:h=1280,format=rgb24", 480, 854),     # scripts/processVideos.js:96
    ("scale=w=720:h=1280,format=bgr24", 360, 640),
    ("scale=320:568:flags=area,format=rgb24", 1080, 1920),
    ("scale=101:57,format=rgb24", 64, 36),             # odd everything: the last pixel has no pair
    ("scale=64:36,format=rgba", 101, 57),
    ("scale=200:150,crop=101:75:33:21,format=rgb24", 100, 80),      # odd crop offsets: pairs straddle the crop's left edge
    ("scale=200:150,crop=100:75:33:20,pad=131:99:7:5:color=red,format=bgr24", 100, 80),   # pad on RGB: odd sizes, odd origin
    ("crop=64:32:10:6,scale=128:96,format=rgb24", 100, 60),
    ("format=rgb24", 64, 48),                          # conversion only: swscale's unscaled converter
    ("format=bgra", 64, 48),
    ("format=rgb24", 64, 47),                          # odd height: the generic path with identity luma filters
    ("scale=64:48,format=rgb24", 64, 48),              # same-size scale: unscaled converter inside the scale filter
    ("scale=64:48,crop=31:21:5:3,format=rgb24", 64, 48),
    ("crop=32:20,pad=64:48:-1:-1:color=white,format=bgr24", 64, 48),   # yuv crop + pad, converted with the borders
    ("pad=80:60:-1:-1,format=rgb24", 63, 47),          # odd picture in a yuv pad: its last column goes to the fill
    # vertical filters of 1-2 taps on luma AND chroma: vscale.c packed_vscale() takes yuv2packed2 (both 2 taps) or
    # yuv2packed1 (luma 1 tap, chroma 1-2) row by row, not the 19-bit sums of yuv2packedX (round 2 refused these)
    ("scale=128:96:flags=bilinear,format=rgb24", 64, 48),      # both axes enlarged, 2 taps each: yuv2rgb_2_c_template
    ("scale=101:77:flags=bilinear,format=bgra", 64, 48),       # odd sizes, folded edge rows
    ("scale=128:48:flags=bilinear,format=rgb24", 64, 48),      # height kept: luma 1 tap, chroma 2 (24 -> 48 rows): yuv2rgb_1_c_template, chroma from one line or the plain average
    ("scale=128:96:flags=area,format=rgb24", 64, 48),          # area enlarging is swscale's bilinear
    ("scale=96:48,format=rgb24", 64, 48),                      # height kept, bicubic: luma 1 tap, chroma 4: yuv2packedX after all
    # packed RGB from a full-range link: the range pass, then a conversion pass over a yuvj420p source (full-range tables)
    ("format=yuvj420p,format=rgb24", 64, 48),
    ("format=yuvj420p,format=bgra", 64, 47),                   # odd height: the conversion pass is the general scaler
    (THUMB + ",format=yuv420p,format=yuvj420p,format=rgb24", 720, 1280),   # src/services/video.js:329-334's chain as the mjpeg encoder gets it, as pixels
    ("crop=32:20:4:2,pad=64:48:-1:-1:color=white,format=yuvj420p,format=bgr24", 64, 48),
]


def run_case(tts, oracle, vf, iw, ih, patterns, n=2):
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih)
    assert plan.geom.out_fmt != 0 and plan.out_bytes == plan.geom.out_w * plan.geom.out_h * plan.out_bpp
    for pat in patterns:
        src = synth.frames(pat, iw, ih, n)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda())
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d %s [%s]: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, pat, plan.kernel_name(0), bad.size, bad[0], got.ravel()[bad[0]], want.ravel()[bad[0]])
    plan.close()


@pytest.mark.parametrize("vf,iw,ih", CASES)
def test_rgb_chain_matches_oracle(tts, oracle, vf, iw, ih):
    big = iw * ih > 1000 * 1000
    pats = ("random", "impulse") if big else synth.PATTERNS
    run_case(tts, oracle, vf, iw, ih, pats, n=1 if big else 3)


def test_rgb_batch_ragged_tail(tts, oracle):
    run_case(tts, oracle, "scale=96:54,format=rgb24", 64, 36, ("random",), n=37)


def test_rgb_fast_and_generic_kernels_agree(tts, oracle, monkeypatch):
    import torch
    src = synth.frames("random", 320, 180, 3)
    vf = "scale=568:320,format=rgb24"
    plan = tts.Plan.from_vf(vf, 320, 180)
    a = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    monkeypatch.setenv("TTS_FORCE_GENERIC", "1")
    plan2 = tts.Plan.from_vf(vf, 320, 180)
    assert plan2.kernel_name(0).startswith("rgb_generic")
    b = plan2.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(a, b) and np.array_equal(a, oracle.run_chain(plan.geom, src))


SCRATCH_CASES = [
    (THUMB, 1920, 1080, "rgb24"),                      # src/services/video.js:329-334 on 1080p
    (THUMB, 1080, 1920, "rgb24"),                      # 101 x 180 picture in the 320 x 180 canvas: pad border on RGB pixels
    (THUMB, 3840, 2160, "bgra"),
    ("scale=160:90", 1920, 1080, "bgr24"),
    ("scale=213:121", 1283, 721, "rgba"),              # odd picture: the planes only fit a 4-byte-a-pixel frame
    ("scale=214:120,crop=200:100:6:10", 1283, 721, "rgb24"),
    ("scale=96:54,pad=128:72:-1:-1:color=red", 640, 360, "rgb24"),
    ("scale=1280:720", 3840, 2160, "rgb24"),           # no picture-size limit: rows are converted independently
    ("scale=640:360", 1920, 1080, "bgra"),
    ("scale=1280:720", 1920, 1080, "rgb24"),           # 1.5x: 6-7 taps -- the grouped-window strip kernel with the 16-bit luma store
    ("scale=720:1280", 1080, 1920, "bgra"),            # the web compress box (portrait.test.js:70-97) as pixels
    ("scale=300:200", 400, 300, "rgb24"),
    ("scale=1080:608", 1920, 1080, "bgr24"),           # 135 eight-pixel units a row: the last wave's string ends in half a chunk
    ("scale=200:120,pad=240:136:-1:-1:color=white", 640, 360, "rgb24"),   # 25 units
]


UNSCALED_CASES = [
    ("format=rgb24", 1920, 1080, "rgb_unscaled8"),            # the streaming form: 8 pixels of a row pair a lane
    ("format=bgra", 1920, 1080, "rgb_unscaled8"),
    ("format=bgr24", 1280, 720, "rgb_unscaled8"),
    ("format=rgba", 48, 32, "rgb_unscaled8"),                 # one partial wave a row pair
    ("format=rgb24", 1040, 64, "rgb_unscaled8"),              # 130 units: the third wave of a row pair holds two
    ("crop=640:360:16:10,format=rgb24", 1280, 720, "rgb_unscaled8"),
    ("crop=640:360:4:10,format=rgb24", 1280, 720, "rgb_unscaled"),     # source rows off the dword grid
    ("format=rgb24", 1080, 1920, "rgb_unscaled8"),            # portrait frames: 135 units a row, the third wave of a row pair ends in half a chunk
    ("format=bgr24", 1928, 1080, "rgb_unscaled8"),            # a multiple of 8, not of 16
    ("format=rgba", 1080, 64, "rgb_unscaled8"),
    ("format=rgb24", 24, 16, "rgb_unscaled8"),                # three units: one and a half chunks... of a single wave
    ("format=rgb24", 1924, 1080, "rgb_unscaled"),             # not a multiple of 8 wide
    ("pad=1936:1096:-1:-1,format=rgb24", 1920, 1080, "rgb_unscaled"),
]


@pytest.mark.parametrize("vf,iw,ih,kernel", UNSCALED_CASES)
def test_rgb_conversion_alone_streaming_kernel(tts, oracle, monkeypatch, vf, iw, ih, kernel):
    """`format=rgb24` with no scale: the 8-pixels-a-lane kernel where the geometry fits, the per-pixel one elsewhere;
    both give the oracle's bytes (TTS_RGB_CONVERT4=1 forces the per-pixel kernel)."""
    import torch
    n = 3
    plan = tts.Plan.from_vf(vf, iw, ih)
    for kind in ("random", "impulse"):
        src = synth.frames(kind, iw, ih, n)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s: %d bytes differ, first at %d (got %d want %d)" % (vf, kind, bad.size, bad[0], got.ravel()[bad[0]], want.ravel()[bad[0]])
    assert plan.kernel_name(0) == kernel, plan.kernel_name(0)
    monkeypatch.setenv("TTS_RGB_CONVERT4", "1")
    slow = tts.Plan.from_vf(vf, iw, ih)
    src = synth.frames("random", iw, ih, n)
    d = torch.from_numpy(src).cuda()
    assert np.array_equal(slow.run_batch(d).cpu().numpy(), plan.run_batch(d).cpu().numpy())


STRIP16_CASES = [
    ("scale=1280:720", 1920, 1080, "rgb24"),           # 1.5x: four columns per 12-byte window
    ("scale=720:1280", 1080, 1920, "bgra"),
    ("scale=1600:900", 1920, 1080, "rgb24"),           # 1.2x: 5-6 taps
    ("scale=1120:630", 1920, 1080, "bgr24"),           # 1.71x: 7-8 taps, two columns per window
    ("scale=1920:1080", 3840, 2160, "rgb24"),          # 2x: 8 taps, two DMAs a row, four rows in flight
    ("scale=1100:1000", 1920, 1080, "rgba"),           # 1.75x across, 1.08x down: 8-byte windows with a 6-row ring
    ("scale=1700:600,crop=1500:500:100:50", 1920, 1080, "rgb24"),
]


@pytest.mark.parametrize("vf,iw,ih,fmt", STRIP16_CASES)
def test_rgb_small_shrinks_strip_kernel_with_16_bit_luma(tts, oracle, monkeypatch, vf, iw, ih, fmt):
    """5-8 taps an axis without the fused walk (TTS_NO_RGB_WALK=1: what per-frame pointers still run): the scratch planes
    come from the LDS strip kernels (their Y16 instantiations) instead of the column walk; bytes as the oracle's."""
    import torch
    chain = vf + ",format=" + fmt
    monkeypatch.setenv("TTS_NO_RGB_WALK", "1")
    fast = tts.Plan.from_vf(chain, iw, ih)
    name = fast.kernel_name(0)
    assert name.startswith("strip_lds") and name.endswith("+rgb_convert"), name
    n = 1 if iw * ih > 4_000_000 else 3
    for kind in ("random", "impulse"):
        src = synth.frames(kind, iw, ih, n)
        got = fast.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        want = oracle.run_chain(fast.geom, src)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s: %d bytes differ, first at %d" % (chain, kind, bad.size, bad[0])


WALK_CASES = STRIP16_CASES + [
    ("scale=720:1280", 1080, 1920, "rgb24"),           # the web compress box (portrait.test.js:70-97) as pixels
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 3840, 2160, "rgb24"),   # 4K -> portrait: 1.125x, a crop behind the scale
    ("scale=1280:720", 1920, 1080, "bgra"),
    ("scale=644:362", 1280, 720, "rgb24"),             # not a multiple of 8 wide: the last strip slid left, no sector exchange
    ("scale=330:250", 400, 300, "bgr24"),              # a wave's strips wrap over several frames
    ("scale=1278:718,crop=1000:600:100:50", 1920, 1080, "rgba"),
    ("scale=960:1080", 1920, 1080, "rgb24"),           # 2x across (8 taps), nothing down the picture: one-tap luma rows, two-tap chroma
    ("scale=1920:600", 1920, 1080, "rgb24"),           # nothing across, 1.8x down
    ("scale=1280:720:flags=bilinear", 1920, 1080, "rgb24"),
    ("scale=1280:720:flags=area", 1920, 1080, "rgb24"),
]


@pytest.mark.parametrize("vf,iw,ih,fmt", WALK_CASES)
def test_rgb_small_shrinks_fused_walk(tts, oracle, vf, iw, ih, fmt):
    """5-8 taps an axis, packed batches: ONE kernel filters luma and chroma side by side and writes the pixels once
    (rgb_walk_kernel); bytes as the oracle's, batches whose waves wrap over frames included."""
    import torch
    chain = vf + ",format=" + fmt
    plan = tts.Plan.from_vf(chain, iw, ih)
    name = plan.kernel_name(0)
    n = 1 if iw * ih > 4_000_000 else 5
    for kind in ("random", "impulse"):
        src = synth.frames(kind, iw, ih, n)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        want = oracle.run_chain(plan.geom, src)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s [%s]: %d bytes differ, first at %d (got %d want %d)" % (chain, kind, name, bad.size, bad[0], got.ravel()[bad[0]], want.ravel()[bad[0]])
    if "flags" not in vf and "960:1080" not in vf and "1920:600" not in vf:
        assert name.startswith("rgb_walk<"), name


@pytest.mark.parametrize("vf,iw,ih,fmt", SCRATCH_CASES)
def test_rgb_thumbnails_column_walk_and_in_place_conversion(tts, oracle, monkeypatch, vf, iw, ih, fmt):
    """Many-tap shrinks written as packed RGB: the planar column walk leaves unclipped 16-bit luma and chroma at
    ceil(w / 2) x h in the destination frame, one workgroup per frame converts in place -- same bytes as the
    one-lane-per-pixel kernel (TTS_NO_RGB_SCRATCH=1) and as the oracle, for batches too."""
    import torch
    chain = vf + ",format=" + fmt
    n = 1 if iw * ih > 4_000_000 else 5
    src = synth.frames("random", iw, ih, n)
    fast = tts.Plan.from_vf(chain, iw, ih)
    assert fast.kernel_name(0).endswith("+rgb_convert") or fast.kernel_name(0).startswith("rgb_walk<"), fast.kernel_name(0)   # (5-8 taps: the fused walk)
    monkeypatch.setenv("TTS_NO_RGB_SCRATCH", "1")
    monkeypatch.setenv("TTS_NO_RGB_WALK", "1")
    slow = tts.Plan.from_vf(chain, iw, ih)
    assert slow.kernel_name(0).startswith("rgb_generic"), slow.kernel_name(0)
    d = torch.from_numpy(src).cuda()
    a, b = fast.run_batch(d).cpu().numpy(), slow.run_batch(d).cpu().numpy()
    want = oracle.run_chain(fast.geom, src)
    assert np.array_equal(b, want), "generic kernel differs from the oracle"
    bad = np.flatnonzero(a != want)
    assert bad.size == 0, "%s: %d bytes differ, first at %d (got %d want %d)" % (chain, bad.size, bad[0], a.ravel()[bad[0]], want.ravel()[bad[0]])
    src = synth.frames("impulse", iw, ih, 1)
    assert np.array_equal(fast.run_batch(torch.from_numpy(src).cuda()).cpu().numpy(), oracle.run_chain(fast.geom, src))


def test_rgb_thumbnails_padded_rows(tts, oracle):
    """The same path into rows with padding (tts_run_batch_strided): the scratch of a row lives in the bytes of its
    own pixels, so the padding is never written."""
    import torch
    iw, ih, n = 1280, 720, 4
    plan = tts.Plan.from_vf(THUMB + ",format=rgb24", iw, ih)
    assert plan.kernel_name(0).endswith("+rgb_convert")
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src).reshape(n, oh, ow * 3)
    dstride = ow * 3 + 11
    d = torch.full((n, oh * dstride), 0xA5, dtype=torch.uint8, device="cuda")
    plan.run_batch_strided(torch.from_numpy(src).cuda(), (iw, (iw + 1) // 2), d, (dstride, 0), n)
    got = d.cpu().numpy().reshape(n, oh, dstride)
    assert np.array_equal(got[:, :, : ow * 3], want) and (got[:, :, ow * 3:] == 0xA5).all()


@pytest.mark.parametrize("vf,iw,ih,fmt", [(THUMB, 1280, 720, "rgb24"), (THUMB, 720, 1280, "bgra"), ("scale=40:30", 200, 150, "rgb24")])
def test_rgb_thumbnails_through_frame_pointers(tts, oracle, vf, iw, ih, fmt):
    """tts_run: every source plane and every RGB frame its own allocation, padded RGB rows.  (The last case has planes
    too small for the column walk's per-frame-pointer mode: it must fall back, not break.)"""
    import torch
    n, pad = 4, 9
    bpp = 4 if fmt.endswith("a") else 3
    plan = tts.Plan.from_vf(vf + ",format=" + fmt, iw, ih)
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src).reshape(n, oh, ow * bpp)
    cw = (iw + 1) // 2
    # (a frame's planes are carved out of one allocation, in scrambled order with gaps: packed-RGB output wants the
    # planes of a FRAME within 4 GiB of each other, which separate allocations do not promise)
    outs, sp, dp, ss, ds = [], [], [], [], []
    for i in range(n):
        planes = synth.split(src[i], iw, ih)
        sizes = [p.size for p in planes] + [oh * (ow * bpp + pad)]
        arena = torch.full((sum(sizes) + 4 * 333,), 0x5A, dtype=torch.uint8, device="cuda")
        offs, at = {}, 0
        for k in (2, 3, 0, 1):
            offs[k] = at + 333 - (i % 3)
            at = offs[k] + sizes[k]
        for k in range(3):
            arena[offs[k]: offs[k] + sizes[k]] = torch.from_numpy(np.ascontiguousarray(planes[k]).ravel()).cuda()
        outs.append((arena, offs[3], sizes[3]))
        sp.append([arena.data_ptr() + offs[k] for k in range(3)]); dp.append([arena.data_ptr() + offs[3], 0, 0])
        ss.append([iw, cw, cw]); ds.append([ow * bpp + pad, 0, 0])
    plan.run_frames(sp, dp, ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        arena, o, sz = outs[i]
        got = arena[o: o + sz].cpu().numpy().reshape(oh, ow * bpp + pad)
        assert np.array_equal(got[:, : ow * bpp], want[i]), (vf, fmt, i)
        assert (got[:, ow * bpp:] == 0x5A).all(), "wrote into the stride padding"


@pytest.mark.parametrize("vf,iw,ih,n", [("scale=20:12", 60, 40, 1), ("scale=20:12", 60, 40, 9), ("scale=16:10", 48, 30, 2), ("scale=24:16", 100, 64, 1)])
def test_rgb_tiny_shrinks_fall_back(tts, oracle, vf, iw, ih, n):
    """Sources under 4 KiB run the one-lane-per-sample kernels (no 16-bit luma store there): the RGB thumbnail
    path must not send them through the scratch pass."""
    import torch
    plan = tts.Plan.from_vf(vf + ",format=rgb24", iw, ih)
    src = synth.frames("random", iw, ih, n)
    got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.run_chain(plan.geom, src)), plan.kernel_name(0)


def test_rgb_strided_rows_and_frame_pointers(tts, oracle):
    """Padded RGB rows (tts_run_batch_strided) and per-frame pointers (tts_run: plane[0] / stride[0] alone)."""
    import torch
    iw, ih, n = 160, 90, 5
    plan = tts.Plan.from_vf("scale=284:160,format=rgb24", iw, ih)
    ow, oh = plan.out_size
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src).reshape(n, oh, ow * 3)
    dstride = ow * 3 + 13
    d = torch.full((n, oh * dstride), 0xA5, dtype=torch.uint8, device="cuda")
    plan.run_batch_strided(torch.from_numpy(src).cuda(), (iw, (iw + 1) // 2), d, (dstride, 0), n)
    got = d.cpu().numpy().reshape(n, oh, dstride)
    assert np.array_equal(got[:, :, : ow * 3], want) and (got[:, :, ow * 3:] == 0xA5).all()
    # every plane its own allocation
    cw, ch = (iw + 1) // 2, (ih + 1) // 2
    planes, outs, sp, dp, ss, ds = [], [], [], [], [], []
    for i in range(n):
        y, u, v = synth.split(src[i], iw, ih)
        t = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (y, u, v)]
        o = torch.zeros(oh * ow * 3, dtype=torch.uint8, device="cuda")
        planes.append(t); outs.append(o)
        sp.append([x.data_ptr() for x in t]); dp.append([o.data_ptr(), 0, 0])
        ss.append([iw, cw, cw]); ds.append([ow * 3, 0, 0])
    plan.run_frames(sp, dp, ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        assert np.array_equal(outs[i].cpu().numpy().reshape(oh, ow * 3), want[i])


def test_rgb_unsupported_paths_say_so(tts):
    # the unscaled converter on an odd width and an even height: swscale's yuv2rgb_c_24_rgb never writes the last column
    # (its loops go 8, 4, 2 pixels at a time), so the reference result is whatever the frame buffer held: nothing to match
    with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
        tts.Plan.from_vf("format=rgb24", 63, 48)
    # a range step inside the scaler that writes RGB (yuvj pinned BEFORE the scale) is not a chain libavfilter builds this way
    with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
        tts.Plan.from_vf("format=yuvj420p,scale=128:96,format=rgb24", 64, 48)


# ---- the other direction: packed rgb24 in, yuv420p out ---------------------------------------------------
RGBIN_CASES = [
    ("scale=w=720:h=1280", 480, 854),                  # scripts/processVideos.js:96 fed from an RGB source
    (THUMB + ",format=yuv420p", 720, 1280),            # src/services/video.js:329-332: scale, then pad on yuv420p
    (THUMB + ",format=yuv420p", 1280, 720),
    (PORTRAIT, 1920, 1080),
    (PORTRAIT, 1280, 720),
    ("scale=320:568:flags=area", 1080, 1920),
    ("scale=101:57", 64, 36),
    ("scale=64:36", 102, 57),
    ("crop=64:33:11:7,scale=128:96", 100, 60),         # a crop on RGB pixels: odd offsets, odd height
    ("format=yuv420p", 64, 48),                        # conversion only
    ("crop=32:21:5:3", 64, 48),                        # crop on RGB, then the conversion pass (odd height out)
    ("scale=200:150,crop=100:74:34:20,pad=132:100:8:6:color=red", 100, 80),
    ("scale=96:54", 63, 36),                           # an odd source width: fine where the picture widens (chroma is read per pixel there, not per pair)
    ("scale=64:80", 64, 36),                           # width kept, height enlarged: chroma stays on pixel pairs
    ("scale=264:200", 131, 99),                        # odd width, widened: off the strip walk's grid (its lanes convert pixel pairs)
]


RGBIN_WALK_CASES = [
    ("format=yuv420p", 1920, 1080),                    # the conversion alone: north_star's "YUV420<->RGB colour conversion", RGB in
    ("format=yuv420p", 1280, 720),
    ("format=yuv420p", 64, 48),
    ("format=yuv420p", 72, 35),                        # nine units (one partial wave), odd height: 18 chroma rows out
    ("format=yuv420p", 520, 8),                        # two waves across, the shortest picture the walk takes
    ("scale=640:360", 640, 360),                       # a same-size scale is no scale
    ("crop=320:200:8:6", 640, 360),                    # a crop in front of the conversion: the walk starts inside the frame
]


@pytest.mark.parametrize("vf,iw,ih", RGBIN_WALK_CASES)
def test_rgb24_input_conversion_alone_streaming_walk(tts, oracle, monkeypatch, vf, iw, ih):
    """rgb24 -> yuv420p with nothing resized: rgbin_walk_kernel (8 pixel columns a lane down a row segment, chroma's
    2 : 1 vertical bicubic from a register ring); the tiled kernel (TTS_NO_RGBIN_WALK=1) and the oracle give the same
    bytes; batches long enough for several row segments and single frames."""
    import torch
    walk = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert walk.kernel_name(0).startswith("rgbin_walk"), walk.kernel_name(0)
    monkeypatch.setenv("TTS_NO_RGBIN_WALK", "1")
    tiled = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert not tiled.kernel_name(0).startswith("rgbin_walk"), tiled.kernel_name(0)
    big = iw * ih > 1000 * 1000
    for pat, n in (("random", 1), ("impulse", 2), ("random", 2 if big else 40)):
        src = rgb_frames(pat, iw, ih, n)
        d = torch.from_numpy(src).cuda()
        got = walk.run_batch(d).cpu().numpy()
        want = oracle.run_chain(walk.geom, src[: min(n, 3)])
        bad = np.flatnonzero(got[: min(n, 3)] != want)
        assert bad.size == 0, "%s %dx%d %s n=%d: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, pat, n, bad.size, bad[0], got.ravel()[bad[0]], want.ravel()[bad[0]])
        assert np.array_equal(got, tiled.run_batch(d).cpu().numpy())


RGBIN_STRIP_CASES = [
    (PORTRAIT, 1920, 1080),                            # the portrait chain from an RGB source: 1.78x, crop inside the scaled picture
    (PORTRAIT, 1280, 720),                             # 2.67x
    ("scale=w=720:h=1280", 480, 854),                  # scripts/processVideos.js:96: 1.5x
    ("scale=1280:720", 640, 360),                      # 2x exactly
    ("scale=168:96", 64, 36),                          # a partial wave, odd ratios per axis
    ("scale=720:1280:flags=bilinear", 480, 854),       # 2 taps: windows padded to four
    ("scale=1000:700,crop=800:500:100:101", 640, 360), # crop at an odd row inside the enlargement
    ("scale=1096:618", 1000, 562),                     # 1.096x: chroma shrinks 1.82 : 1 across and down (8 taps), 8 pixels a lane
    ("scale=264:200", 130, 99),                        # an odd source height
]


@pytest.mark.parametrize("vf,iw,ih", RGBIN_STRIP_CASES)
def test_rgb24_input_enlargements_strip_walk(tts, oracle, monkeypatch, vf, iw, ih):
    """rgb24 in, widened: rgbin_strip_kernel (lanes are (frame, strip) pairs; the wave converts its runs of pixels once
    per source row into LDS -- chroma per pixel: rgb24ToUV_c -- and lanes filter 8 luma + 4 + 4 chroma columns from there
    into register rings); the tiled kernel (TTS_NO_RGBIN_STRIP=1) and the oracle give the same bytes; one frame and
    batches whose waves straddle frames, cut into several row segments."""
    import torch
    strip = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert strip.kernel_name(0).startswith("rgbin_strip"), strip.kernel_name(0)
    monkeypatch.setenv("TTS_NO_RGBIN_STRIP", "1")
    tiled = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert tiled.kernel_name(0).startswith("rgbin_tile"), tiled.kernel_name(0)
    big = iw * ih > 500 * 1000
    for pat, n in (("random", 1), ("impulse", 2), ("zone", 2 if big else 24)):
        src = rgb_frames(pat, iw, ih, n)
        d = torch.from_numpy(src).cuda()
        got = strip.run_batch(d).cpu().numpy()
        m = min(n, 2)
        want = oracle.run_chain(strip.geom, src[:m])
        bad = np.flatnonzero(got[:m].ravel() != want.ravel())
        assert bad.size == 0, "%s %dx%d %s n=%d: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, pat, n, bad.size, bad[0], got[:m].ravel()[bad[0]], want.ravel()[bad[0]])
        assert np.array_equal(got, tiled.run_batch(d).cpu().numpy())


RGBIN_SHRINK_CASES = [
    ("scale=960:540", 1920, 1080),                     # 2x exactly: 8 taps across, 16 down the chroma planes
    ("scale=720:1280", 1080, 1920),                    # the web compress box (portrait.test.js:70-97) from an RGB source: 1.5x
    ("scale=1280:720", 1920, 1080),
    ("scale=640:360", 1280, 720),
    ("scale=320:568:flags=area", 1080, 1920),          # BASELINE's thumbnails from an RGB source: 3.4x, 5 taps across (area)
    ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 3840, 2160),   # 4K -> portrait: 1.125x, a crop behind the scale
    ("scale=960:540,crop=800:400:80:70", 1920, 1080),  # crop at an odd row
    ("scale=48:26", 64, 36),                           # a partial wave; 13 chroma rows out
    ("scale=1280:300", 1920, 1080),                    # 1.5x across, 3.6x down: 14 luma and 29 chroma taps down the picture
    ("scale=1718:1000", 1920, 1080),                   # 1.12x: 4-pixel lanes, a width of 4 k + 2
    ("scale=640:360:flags=bilinear", 1280, 720),
]


@pytest.mark.parametrize("vf,iw,ih", RGBIN_SHRINK_CASES)
def test_rgb24_input_shrinks_source_row_walk(tts, oracle, monkeypatch, vf, iw, ih):
    """rgb24 in, shrunk (at most 8 taps across): rgbin_shrink_kernel (lanes are (frame, strip) pairs; the wave converts
    its runs of pixels once per source row into LDS -- chroma per pixel pair: rgb24ToUV_half_c -- lanes filter 4 luma +
    2 + 2 chroma columns from there and add the row into the sums of the output rows that hold it); the tiled kernel
    (TTS_NO_RGBIN_SHRINK=1) and the oracle give the same bytes; one frame and batches whose waves straddle frames."""
    import torch
    fast = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert fast.kernel_name(0).startswith("rgbin_shrink"), fast.kernel_name(0)
    monkeypatch.setenv("TTS_NO_RGBIN_SHRINK", "1")
    tiled = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert not tiled.kernel_name(0).startswith("rgbin_shrink"), tiled.kernel_name(0)
    big = iw * ih > 500 * 1000
    for pat, n in (("random", 1), ("impulse", 2), ("zone", 2 if big else 24)):
        src = rgb_frames(pat, iw, ih, n)
        d = torch.from_numpy(src).cuda()
        got = fast.run_batch(d).cpu().numpy()
        m = min(n, 2)
        want = oracle.run_chain(fast.geom, src[:m])
        bad = np.flatnonzero(got[:m].ravel() != want.ravel())
        assert bad.size == 0, "%s %dx%d %s n=%d [%s]: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, pat, n, fast.kernel_name(0), bad.size, bad[0], got[:m].ravel()[bad[0]], want.ravel()[bad[0]])
        assert np.array_equal(got, tiled.run_batch(d).cpu().numpy())


def rgb_frames(pattern, w, h, n):
    """Packed rgb24 test frames: the three planes of the yuv generator at full size, interleaved."""
    y = synth.frames(pattern, 2 * w, 2 * h, n)         # gives planes of w x h as its chroma
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        a, b, c = synth.split(y[i], 2 * w, 2 * h)
        out[i, :, :, 0] = a[::2, ::2]; out[i, :, :, 1] = b; out[i, :, :, 2] = c
    return out.reshape(n, -1)


def repack(rgb, w, h, fmt):
    """rgb24 frames [n, h * w * 3] in another packed layout (alpha: a byte pattern nothing may depend on)."""
    n = rgb.shape[0]
    px = rgb.reshape(n, h, w, 3)
    if fmt == "rgb24":
        return rgb
    if fmt == "bgr24":
        return np.ascontiguousarray(px[..., ::-1]).reshape(n, -1)
    out = np.empty((n, h, w, 4), np.uint8)
    out[..., :3] = px if fmt == "rgba" else px[..., ::-1]
    out[..., 3] = (np.arange(w, dtype=np.uint32) * 37 + 11).astype(np.uint8)
    return out.reshape(n, -1)


PACKED_IN_CASES = [
    ("format=yuv420p", 640, 360),                      # conversion alone: the streaming walk for the 3-byte layouts
    (PORTRAIT, 1280, 720),                             # an enlargement: the strip walk (bgr24), the tiled kernel (4-byte pixels)
    ("scale=640:360", 1280, 720),                      # a shrink: the source-row walk (bgr24)
    (THUMB + ",format=yuv420p", 720, 1280),            # the reference's thumbnail chain, pad on yuv420p behind the scale
    ("crop=64:33:11:7,scale=128:96", 100, 60),         # a crop on the packed pixels at odd offsets
    ("scale=96:54", 63, 36),                           # an odd source width, widened
]


@pytest.mark.parametrize("fmt", ["bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("vf,iw,ih", PACKED_IN_CASES)
def test_packed_rgb_input_other_layouts(tts, oracle, vf, iw, ih, fmt):
    """`-pix_fmt bgr24 | rgba | bgra` in front of `-i`: the oracle's bytes, and the bytes the rgb24 plan gives for the
    same pixels repacked (alpha is not read)."""
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=fmt)
    ref = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    special = fmt == "bgr24" and vf == "format=yuv420p"    # the same size: swscale's bgr24toYV12 converter, other bytes than the scaler's
    assert plan.in_bytes == iw * ih * (4 if fmt.endswith("a") else 3) and plan.kernel_name(0).startswith("bgr24" if special else "rgbin")
    if special:
        assert plan.kernel_name(0) == "bgr24_yv12_rows"
    elif fmt == "bgr24":
        assert plan.kernel_name(0) == ref.kernel_name(0), (plan.kernel_name(0), ref.kernel_name(0))   # (every walk takes the swapped layout)
    for pat in ("random", "impulse"):
        rgb = rgb_frames(pat, iw, ih, 3)
        src = repack(rgb, iw, ih, fmt)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        want = oracle.run_chain(plan.geom, src)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s %dx%d %s [%s]: %d bytes differ, first at %d" % (fmt, vf, iw, ih, pat, plan.kernel_name(0), bad.size, bad[0])
        assert np.array_equal(got, ref.run_batch(torch.from_numpy(rgb).cuda()).cpu().numpy()) != special


BGR24_SAME_SIZE_CASES = [
    ("format=yuv420p", 1920, 1080),
    ("format=yuv420p", 66, 37),                         # an odd height: the last row is an even row and has chroma
    ("format=yuv420p", 8, 2),
    ("format=yuv420p", 6, 3),                           # narrower than a lane's eight pixels: a lane per sample
    ("format=yuv420p", 1034, 11),                       # 129 and a quarter lanes a row: the last is slid left
    ("crop=64:33:11:7", 100, 60),                       # a crop on the packed pixels, odd offsets, odd height
    ("scale=100:60,crop=64:32:12:8", 100, 60),          # a scale to the same size is still that converter; the crop behind it is planar
    ("scale=100:60,pad=128:96:-1:-1:color=red", 100, 60),
    ("crop=80:40:3:5,scale=80:40,crop=64:32,pad=96:64:-1:-1", 100, 60),
    ("crop=80:40:3:5,format=yuv420p,crop=64:32,pad=96:64:-1:-1", 100, 60),   # format=yuv420p as the conversion point: the same chain
]


@pytest.mark.parametrize("vf,iw,ih", BGR24_SAME_SIZE_CASES)
def test_bgr24_same_size_special_converter(tts, oracle, monkeypatch, vf, iw, ih):
    """bgr24 in, nothing resized: ff_rgb24toyv12_c's bytes (oracle: tto_sws_packed_to_yuv420's first branch), through
    batches and through per-frame pointers; an odd width is refused with the reason."""
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt="bgr24")
    # nothing behind the converter: eight pixels of two rows a lane; a crop or pad behind it: a lane per output sample
    assert plan.kernel_name(0) == ("bgr24_yv12" if "," in vf or iw < 8 else "bgr24_yv12_rows"), plan.kernel_name(0)
    monkeypatch.setenv("TTS_FORCE_GENERIC", "1")
    slow = tts.Plan.from_vf(vf, iw, ih, in_fmt="bgr24")
    monkeypatch.delenv("TTS_FORCE_GENERIC")
    assert slow.kernel_name(0) == "bgr24_yv12"
    for pat in ("random", "impulse", "hramp"):
        src = repack(rgb_frames(pat, iw, ih, 3), iw, ih, "bgr24")
        want = oracle.run_chain(plan.geom, src)
        d = torch.from_numpy(src).cuda()
        got = plan.run_batch(d).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d %s: %d bytes differ, first at %d" % (vf, iw, ih, pat, bad.size, bad[0])
        assert np.array_equal(slow.run_batch(d).cpu().numpy(), want)
    # per-frame pointers, padded strides
    n, ow, oh = 3, plan.out_size[0], plan.out_size[1]
    keep, sp, ss, dp, ds = [], [], [], [], []
    for i in range(n):
        t = torch.zeros((ih, iw * 3 + 5), dtype=torch.uint8, device="cuda")
        t[:, : iw * 3] = torch.from_numpy(src[i].reshape(ih, iw * 3)).cuda()
        outs = [torch.full(((oh + 1) // 2 if k else oh, ((ow + 1) // 2 if k else ow) + 3), 9, dtype=torch.uint8, device="cuda") for k in (0, 1, 1)]
        keep += [t] + outs
        sp.append([t.data_ptr(), 0, 0]); ss.append([t.shape[1], 0, 0])
        dp.append(outs); ds.append([o.shape[1] for o in outs])
    plan.run_frames(sp, [[o.data_ptr() for o in f] for f in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        for k, w in enumerate(synth.split(want[i], ow, oh)):
            g = dp[i][k].cpu().numpy()
            assert np.array_equal(g[:, : w.shape[1]], w), ("per-frame pointers", i, k)
            assert (g[:, w.shape[1]:] == 9).all(), "wrote into the stride padding"
    with pytest.raises(Exception) as e:
        tts.Plan.from_vf("format=yuv420p", iw - 1, ih, in_fmt="bgr24")
    assert "UNSUPPORTED" in str(e.value) or "odd width" in str(e.value), str(e.value)


@pytest.mark.parametrize("vf,iw,ih", RGBIN_CASES)
def test_rgb24_input_matches_oracle(tts, oracle, vf, iw, ih):
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert plan.geom.in_fmt == 1 and plan.in_bytes == iw * ih * 3 and plan.kernel_name(0).startswith("rgbin")
    big = iw * ih > 1000 * 1000
    for pat in (("random", "impulse") if big else synth.PATTERNS):
        src = rgb_frames(pat, iw, ih, 1 if big else 3)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d %s: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, pat, bad.size, bad[0], got.ravel()[bad[0]], want.ravel()[bad[0]])


@pytest.mark.parametrize("vf,iw,ih", [RGBIN_CASES[0], RGBIN_CASES[7], RGBIN_CASES[8], RGBIN_CASES[11]])
def test_rgb24_input_tiled_and_generic_kernels_agree(tts, oracle, monkeypatch, vf, iw, ih):
    """rgbin_tile_kernel (a workgroup converts its tile's source samples once, into LDS) is what runs by default;
    the one-lane-per-sample kernel stays behind it for tiles that do not fit: same bytes from both."""
    import torch
    src = rgb_frames("random", iw, ih, 4)
    monkeypatch.setenv("TTS_NO_RGBIN_STRIP", "1")        # (enlargements take the strip walk by default: its own test below)
    monkeypatch.setenv("TTS_NO_RGBIN_SHRINK", "1")       # (shrinks the source-row walk)
    tiled = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert tiled.kernel_name(0).startswith("rgbin_tile"), tiled.kernel_name(0)
    monkeypatch.setenv("TTS_FORCE_GENERIC", "1")
    generic = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    assert generic.kernel_name(0).startswith("rgbin_generic"), generic.kernel_name(0)
    d = torch.from_numpy(src).cuda()
    a, b = tiled.run_batch(d).cpu().numpy(), generic.run_batch(d).cpu().numpy()
    assert np.array_equal(a, b) and np.array_equal(a, oracle.run_chain(tiled.geom, src))


@pytest.mark.parametrize("planewise,pad_s,pad_d", [("0", 7, 5), ("1", 7, 5), ("0", 9, 8)])
@pytest.mark.parametrize("vf,iw,ih", [RGBIN_CASES[0], RGBIN_CASES[8], RGBIN_CASES[11], RGBIN_CASES[2]])
def test_rgb24_input_through_frame_pointers(tts, oracle, monkeypatch, vf, iw, ih, planewise, pad_s, pad_d):
    """tts_run with packed-RGB source frames: one source plane per frame (rows padded), three destination planes,
    every buffer its own allocation; padding is never written.  (Destination rows padded by 8 stay on the grid the
    strip walk's stores need: enlargements take it through the pointer tables too; by 5 they fall back.)"""
    import torch
    n = 5
    monkeypatch.setenv("TTS_PLANEWISE", planewise)       # "1": one destination plane per launch (planes further than 4 GiB apart)
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
    ow, oh = plan.out_size
    src = rgb_frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    keep, sp, ss, dp, ds = [], [], [], [], []
    for i in range(n):
        t = torch.zeros((ih, iw * 3 + pad_s), dtype=torch.uint8, device="cuda")
        t[:, : iw * 3] = torch.from_numpy(src[i].reshape(ih, iw * 3)).cuda()
        outs = [torch.full(((oh + 1) // 2 if k else oh, ((ow + 1) // 2 if k else ow) + pad_d), 9, dtype=torch.uint8, device="cuda") for k in (0, 1, 1)]
        keep += [t] + outs
        sp.append([t.data_ptr(), 0, 0]); ss.append([t.shape[1], 0, 0])
        dp.append(outs); ds.append([o.shape[1] for o in outs])
    plan.run_frames(sp, [[o.data_ptr() for o in f] for f in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        for k, w in enumerate(synth.split(want[i], ow, oh)):
            g = dp[i][k].cpu().numpy()
            assert np.array_equal(g[:, : w.shape[1]], w), (vf, i, k)
            assert (g[:, w.shape[1]:] == 9).all(), "wrote into the stride padding"


@pytest.mark.parametrize("pad_s,pad_d", [(16, 32), (7, 5)])
def test_conversion_alone_through_frame_pointers(tts, oracle, pad_s, pad_d):
    """tts_run, conversion only, both directions: planes of their own, rows padded.  With padding that keeps rows on
    the grid the streaming kernels need (16, 32) they run; with (7, 5) the general kernels do; same bytes, padding
    untouched."""
    import torch
    iw, ih, n = 320, 64, 4
    # yuv420p -> rgb24
    plan = tts.Plan.from_vf("format=rgb24", iw, ih)
    src = synth.frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src).reshape(n, ih, iw * 3)
    keep, sp, ss, dp, ds = [], [], [], [], []
    for i in range(n):
        planes, strides = [], []
        for pl in synth.split(src[i], iw, ih):
            t = torch.zeros((pl.shape[0], pl.shape[1] + pad_s), dtype=torch.uint8, device="cuda")
            t[:, : pl.shape[1]] = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
            keep.append(t); planes.append(t.data_ptr()); strides.append(t.shape[1])
        o = torch.full((ih, iw * 3 + pad_d), 9, dtype=torch.uint8, device="cuda")
        keep.append(o)
        sp.append(planes); ss.append(strides); dp.append(o); ds.append([o.shape[1], 0, 0])
    plan.run_frames(sp, [[o.data_ptr(), 0, 0] for o in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        g = dp[i].cpu().numpy()
        assert np.array_equal(g[:, : iw * 3], want[i]), ("yuv -> rgb", i)
        assert (g[:, iw * 3:] == 9).all(), "wrote into the stride padding"
    # rgb24 -> yuv420p
    plan = tts.Plan.from_vf("format=yuv420p", iw, ih, in_fmt="rgb24")
    src = rgb_frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    keep, sp, ss, dp, ds = [], [], [], [], []
    for i in range(n):
        t = torch.zeros((ih, iw * 3 + pad_s), dtype=torch.uint8, device="cuda")
        t[:, : iw * 3] = torch.from_numpy(src[i].reshape(ih, iw * 3)).cuda()
        outs = [torch.full((ih // 2 if k else ih, (iw // 2 if k else iw) + pad_d), 9, dtype=torch.uint8, device="cuda") for k in (0, 1, 1)]
        keep += [t] + outs
        sp.append([t.data_ptr(), 0, 0]); ss.append([t.shape[1], 0, 0])
        dp.append(outs); ds.append([o.shape[1] for o in outs])
    plan.run_frames(sp, [[o.data_ptr() for o in f] for f in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        for k, w in enumerate(synth.split(want[i], iw, ih)):
            g = dp[i][k].cpu().numpy()
            assert np.array_equal(g[:, : w.shape[1]], w), ("rgb -> yuv", i, k)
            assert (g[:, w.shape[1]:] == 9).all(), "wrote into the stride padding"


def test_rgb24_input_host_path_and_rejections(tts, oracle):
    plan = tts.Plan.from_vf("scale=96:54", 64, 36, in_fmt="rgb24")
    src = rgb_frames("zone", 64, 36, 5)
    assert np.array_equal(plan.run_host(src), oracle.run_chain(plan.geom, src))
    for vf, iw in (("scale=96:54,format=rgb24", 64), ("scale=48:27", 63), ("format=yuv420p", 63), ("pad=81:60", 64), ("scale=96:54,format=yuv420p,format=yuvj420p,format=yuv420p", 64)):
        with pytest.raises(tts.TtsError, match="UNSUPPORTED"):
            tts.Plan.from_vf(vf, iw, 36, in_fmt="rgb24")


RGBIN_PAD_CASES = [
    ("pad=80:60:7:5:color=red", 64, 36),                # an odd offset: pixel pairs straddle the picture's edges
    ("pad=320:180:-1:-1:color=black", 102, 180),        # the reference's letterbox (src/services/video.js:331) on RGB pixels
    ("crop=31:20:3:1,pad=80:61:-1:-1:color=0x336699", 64, 36),   # an odd picture width and an odd canvas height
    ("pad=iw+2:ih+2:1:1:color=white", 638, 358),
]


@pytest.mark.parametrize("fmt", ["rgb24", "bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("vf,iw,ih", RGBIN_PAD_CASES)
def test_packed_rgb_input_pad_without_scale(tts, oracle, vf, iw, ih, fmt):
    """No scale in the chain: the pad runs on RGB pixels and the conversion pass libavfilter puts behind it reads the canvas
    (oracle: tto_run_chain_rgbin, rgb_stage 2).  The kernels never store that canvas: they read through its rectangle."""
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=fmt)
    assert plan.geom.rgb_stage == 2 and plan.kernel_name(0).endswith("+rgb_pad"), plan.kernel_name(0)
    assert plan.kernel_name(0).startswith("bgr24_yv12" if fmt == "bgr24" else "rgbin_generic")
    for pat in ("random", "impulse"):
        src = repack(rgb_frames(pat, iw, ih, 3), iw, ih, fmt)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s %dx%d %s: %d bytes differ, first at %d" % (fmt, vf, iw, ih, pat, bad.size, bad[0])
    # the same picture behind a scale to its own size is converted first and padded on yuv420p: a different chain
    if "crop" not in vf:
        other = tts.Plan.from_vf("scale=%d:%d,%s" % (iw, ih, vf), iw, ih, in_fmt=fmt)
        assert other.geom.rgb_stage == 0 and not other.kernel_name(0).endswith("+rgb_pad")


RGBIN_YUVJ_CASES = [
    ("format=yuvj420p", 64, 36),                                         # the inserted conversion pass writes full range itself
    ("format=yuvj420p", 1280, 720),                                      # ... at a size whose batches span row segments (rgbin_walk<..>+range15 for 3-byte pixels)
    ("scale=96:54,format=yuvj420p", 64, 36),                             # ... or the scale does (an enlargement: chroma per pixel)
    ("scale=640:360,format=yuvj420p", 1280, 720),                        # a shrink (chroma from pixel pairs)
    ("crop=100:60:7:3,scale=50:30,pad=64:40:-1:-1:color=red,format=yuvj420p", 128, 72),   # the pad behind it on a yuvj420p link
    ("pad=80:60:7:5:color=red,format=yuvj420p", 64, 36),                 # a pad on the RGB pixels in front of it
]


@pytest.mark.parametrize("fmt", ["rgb24", "bgr24", "rgba"])
@pytest.mark.parametrize("vf,iw,ih", RGBIN_YUVJ_CASES)
def test_packed_rgb_input_into_yuvj420p(tts, oracle, vf, iw, ih, fmt):
    """Fed with packed RGB, handing out yuvj420p: the converting context applies lumRangeToJpeg_c / chrRangeToJpeg_c to its 15-bit
    rows (oracle: tto_sws_packed_to_yuv420_range); the general kernel carries that (+range15 in its name)."""
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=fmt)
    assert plan.geom.sc_range == 1 and plan.geom.out_range == 0 and "+range15" in plan.kernel_name(0), plan.kernel_name(0)
    limited = tts.Plan.from_vf(vf.replace(",format=yuvj420p", "").replace("format=yuvj420p", "format=yuv420p"), iw, ih, in_fmt=fmt)
    for pat in ("random", "impulse"):
        src = repack(rgb_frames(pat, iw, ih, 3), iw, ih, fmt)
        want = oracle.run_chain(plan.geom, src)
        d = torch.from_numpy(src).cuda()
        got = plan.run_batch(d).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s %dx%d %s [%s]: %d bytes differ, first at %d" % (fmt, vf, iw, ih, pat, plan.kernel_name(0), bad.size, bad[0])
        assert not np.array_equal(got, limited.run_batch(d).cpu().numpy())


THUMB_PINNED_RGBIN = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p,format=yuvj420p"
RGBIN_RANGE_PASS_CASES = [
    (THUMB_PINNED_RGBIN, 720, 1280, (0, 1)),                              # the reference's thumbnail chain + its mjpeg encoder, fed with RGB frames
    (THUMB_PINNED_RGBIN, 1280, 720, (0, 1)),
    ("format=yuv420p,format=yuvj420p", 64, 36, (0, 1)),                    # no scale: the conversion pass, then the range pass
    ("scale=96:54,format=yuv420p,format=yuvj420p", 64, 36, (0, 1)),        # an enlargement (the strip walk)
    ("scale=96:54,format=yuvj420p,format=yuv420p", 64, 36, (1, 2)),        # full range out of the converting context, back to limited at the end
    ("pad=80:60:7:5:color=red,format=yuv420p,format=yuvj420p", 64, 36, (0, 1)),   # the pad on RGB pixels, conversion, range pass
    ("format=yuv420p,pad=80:60:6:4:color=red,format=yuvj420p", 64, 36, (0, 1)),   # the pad behind the conversion, on yuv420p (limited-range red), range pass
]


@pytest.mark.parametrize("fmt", ["rgb24", "bgr24", "rgba"])
@pytest.mark.parametrize("vf,iw,ih,steps", RGBIN_RANGE_PASS_CASES)
def test_packed_rgb_input_then_a_range_pass(tts, oracle, vf, iw, ih, steps, fmt):
    """Fed with packed RGB, a yuv format of the OTHER range behind the one the converting context wrote: a range pass of its own
    over the finished frame (tts_geom.out_range), border included -- a launch of range_kernel behind the RGB-input kernels."""
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=fmt)
    assert (plan.geom.sc_range, plan.geom.out_range) == steps and plan.kernel_name(0).endswith("+range_pass"), plan.kernel_name(0)
    for pat in ("random", "impulse"):
        src = repack(rgb_frames(pat, iw, ih, 3), iw, ih, fmt)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s %dx%d %s [%s]: %d bytes differ, first at %d" % (fmt, vf, iw, ih, pat, plan.kernel_name(0), bad.size, bad[0])
    # the unpinned form of the same chain converts inside the context: other bytes
    if steps == (0, 1) and "scale" in vf:
        inside = tts.Plan.from_vf(vf.replace(",format=yuv420p,format=yuvj420p", ",format=yuvj420p"), iw, ih, in_fmt=fmt)
        assert (inside.geom.sc_range, inside.geom.out_range) == (1, 0)
        assert not np.array_equal(got, inside.run_batch(torch.from_numpy(src).cuda()).cpu().numpy())


@pytest.mark.parametrize("planewise", [False, True])
def test_packed_rgb_input_range_pass_through_frame_pointers(tts, oracle, monkeypatch, planewise):
    """The same through tts_run: every plane its own allocation, padded rows (the range pass must stay off the padding)."""
    import torch
    if planewise:
        monkeypatch.setenv("TTS_PLANEWISE", "1")
    iw, ih, n = 72, 128, 5
    plan = tts.Plan.from_vf(THUMB_PINNED_RGBIN, iw, ih, in_fmt="rgb24")
    ow, oh = plan.out_size
    src = rgb_frames("random", iw, ih, n)
    want = oracle.run_chain(plan.geom, src)
    sp, dp, ss, ds, keep = [], [], [], [], []
    for i in range(n):
        t = torch.zeros((ih, iw * 3 + 12), dtype=torch.uint8, device="cuda")
        t[:, : iw * 3] = torch.from_numpy(src[i].reshape(ih, iw * 3)).cuda()
        keep.append(t); sp.append([t.data_ptr(), 0, 0]); ss.append([iw * 3 + 12, 0, 0])
        outs, ostr = [], []
        for k in range(3):
            w_, h_ = (ow, oh) if k == 0 else ((ow + 1) // 2, (oh + 1) // 2)
            o = torch.full((h_, w_ + 16), 16, dtype=torch.uint8, device="cuda")   # (16: a value the range pass would change)
            outs.append(o); ostr.append(w_ + 16)
        dp.append(outs); ds.append(ostr)
    plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(n):
        for k, w in enumerate(synth.split(want[i], ow, oh)):
            g = dp[i][k].cpu().numpy()
            assert np.array_equal(g[:, : w.shape[1]], w), (i, k)
            assert (g[:, w.shape[1]:] == 16).all(), "the range pass ran over the stride padding"


THUMB_R2 = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p"
RGBIN_MANY_TAP_CASES = [
    (THUMB_R2, 1920, 1080, "rgb24", 2),                                    # the reference's thumbnail chain fed with RGB frames: 6x, 25 taps across; the pad is the whole canvas
    (THUMB_R2, 720, 1280, "rgb24", 3),                                     # a portrait source: 7.1x, the 100-column picture letterboxed at 108
    (THUMB_R2 + ",format=yuvj420p", 1280, 720, "bgr24", 3),                # 4x + the mjpeg encoder's range pass
    (THUMB_R2, 3840, 2160, "rgb24", 1),                                    # 4K: 12x, 49 taps across
    ("scale=640:360", 1920, 1080, "rgb24", 3),                             # 3x: 13 taps
    ("scale=322:182", 1284, 724, "bgr24", 3),                              # 4x off the grids: an output width of 4 k + 2
    ("scale=480:270,crop=400:201:40:33", 1920, 1080, "rgb24", 2),          # 4x and a crop at odd rows behind it
    ("crop=1000:700:100:53,scale=200:140:flags=bilinear,pad=256:144:28:2:color=red", 1284, 800, "rgb24", 3),   # a crop on RGB pixels in front (the row ends on the walk's grid of 4 pixels), bilinear, a pad at its own offset
    ("scale=160:90:flags=area", 1920, 1080, "rgb24", 2),                   # area at 12x
    (THUMB_R2, 1920, 1080, "rgba", 2),                                     # four-byte pixels: the same walk, the fourth byte skipped
    (THUMB_R2, 720, 1280, "bgra", 3),
    ("scale=640:360", 1920, 1080, "bgra", 2),
]


@pytest.mark.parametrize("pxl", [0, 8, 16])
@pytest.mark.parametrize("vf,iw,ih,fmt,n", RGBIN_MANY_TAP_CASES)
def test_packed_rgb_input_many_tap_shrinks(tts, oracle, monkeypatch, vf, iw, ih, fmt, n, pxl):
    """Shrinks of more than 8 taps across from packed RGB: the source-row walk with two luma columns a lane
    (rgbin_shrink_kernel<PXL, TPH, KP, 2, PB>: 8 or 16 pixels converted a lane, the planner's pick and each one forced), a pad
    behind the scale as a border fill in front of it."""
    import torch
    if pxl:
        monkeypatch.setenv("TTS_RGBIN_SHRINK_PXL", str(pxl))
    plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=fmt)
    name = plan.kernel_name(0)
    assert name.startswith("rgbin_shrink<") and ("%dpx" % pxl in name if pxl else "8px" in name or "16px" in name) and ("+pad" in name) == ("pad" in vf), name
    for pat in ("random", "impulse"):
        src = repack(rgb_frames(pat, iw, ih, n), iw, ih, fmt)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %s %dx%d %s [%s]: %d bytes differ, first at %d" % (fmt, vf, iw, ih, pat, name, bad.size, bad[0])


def test_packed_rgb_input_many_tap_shrink_batches_and_pointers(tts, oracle):
    """A batch whose waves straddle frames (40 frames of a 7x portrait thumbnail: 50 strips a frame), against the oracle on a
    few frames and frame against frame on the rest; then the same through per-frame pointers with padded rows."""
    import torch
    iw, ih, n = 360, 640, 40
    plan = tts.Plan.from_vf(THUMB_R2, iw, ih, in_fmt="rgb24")
    assert plan.kernel_name(0).startswith("rgbin_shrink<") and plan.kernel_name(0).endswith("+pad"), plan.kernel_name(0)
    src = rgb_frames("random", iw, ih, n)
    got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
    want = oracle.run_chain(plan.geom, src[:3])
    assert np.array_equal(got[:3], want)
    for k in (3, 17, 39):
        assert np.array_equal(got[k], plan.run_batch(torch.from_numpy(src[k: k + 1]).cuda()).cpu().numpy()[0]), k
    ow, oh = plan.out_size
    sp, dp, ss, ds, keep = [], [], [], [], []
    for i in range(5):
        t = torch.zeros((ih, iw * 3 + 12), dtype=torch.uint8, device="cuda")
        t[:, : iw * 3] = torch.from_numpy(src[i].reshape(ih, iw * 3)).cuda()
        keep.append(t); sp.append([t.data_ptr(), 0, 0]); ss.append([iw * 3 + 12, 0, 0])
        outs, ostr = [], []
        for k in range(3):
            w_, h_ = (ow, oh) if k == 0 else ((ow + 1) // 2, (oh + 1) // 2)
            o = torch.full((h_, w_ + 16), 9, dtype=torch.uint8, device="cuda")
            outs.append(o); ostr.append(w_ + 16)
        dp.append(outs); ds.append(ostr)
    plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
    torch.cuda.synchronize()
    for i in range(5):
        for k, w in enumerate(synth.split(got[i], ow, oh)):
            g = dp[i][k].cpu().numpy()
            assert np.array_equal(g[:, : w.shape[1]], w), (i, k)
            assert (g[:, w.shape[1]:] == 9).all(), "wrote into the stride padding"


def test_packed_rgb_input_thumbnail_chain_host_path(tts, oracle):
    """tts_run_host (pinned staging, chunks): the reference's thumbnail chain + its encoder's range pass fed with host RGB frames --
    border fill, many-tap walk and range pass, chunk after chunk."""
    iw, ih, n = 720, 1280, 7
    plan = tts.Plan.from_vf(THUMB_PINNED_RGBIN, iw, ih, in_fmt="rgb24")
    assert plan.kernel_name(0).startswith("rgbin_shrink<") and plan.kernel_name(0).endswith("+pad+range_pass"), plan.kernel_name(0)
    src = rgb_frames("zone", iw, ih, n)
    assert np.array_equal(plan.run_host(src), oracle.run_chain(plan.geom, src))
