"""Randomised geometries through every kernel family, batch and pointer APIs, against the oracle.
Seeded, so a failure names a reproducible case.  Sizes are kept small (the oracle is scalar C)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _case(rng):
    iw, ih = int(rng.integers(8, 420)), int(rng.integers(8, 300))
    kind = rng.choice(["up", "mild", "down", "deep", "mixed"])
    def axis(n):
        if kind == "up":
            return int(n * rng.uniform(1.0, 3.0))
        if kind == "mild":
            return int(n / rng.uniform(1.0, 2.0))
        if kind == "down":
            return int(n / rng.uniform(2.0, 6.0))
        if kind == "deep":
            return int(n / rng.uniform(6.0, 20.0))
        return int(n * rng.choice([rng.uniform(0.1, 1.0), rng.uniform(1.0, 2.5)]))
    ow, oh = max(2, axis(iw)), max(2, axis(ih))
    flags = rng.choice(["", ":flags=area", ":flags=bilinear", ""])
    vf = "scale=%d:%d%s" % (ow, oh, flags)
    if rng.random() < 0.25 and iw > 40 and ih > 40:          # crop before the scale
        cw, chh = int(rng.integers(16, iw)), int(rng.integers(16, ih))
        vf = "crop=%d:%d:%d:%d," % (cw, chh, int(rng.integers(0, iw - cw + 1)), int(rng.integers(0, ih - chh + 1))) + vf
    if rng.random() < 0.3 and ow > 12 and oh > 12:           # crop after it
        vf += ",crop=%d:%d" % (int(rng.integers(8, ow)), int(rng.integers(8, oh)))
    if rng.random() < 0.2:
        vf += ",pad=iw+%d:ih+%d:-1:-1:color=black" % (2 * int(rng.integers(0, 20)), 2 * int(rng.integers(0, 20)))
    if rng.random() < 0.15:                                  # the range step: inside the scale's context, or pinned to a pass of its own
        vf += str(rng.choice([",format=yuvj420p", ",format=yuv420p,format=yuvj420p"]))
    return vf, iw, ih, int(rng.choice([1, 2, 3, 7, 19, 70]))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_SEEDS", "24"))))
def test_random_geometries(tts, oracle, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    seen = set()
    for _ in range(16):
        vf, iw, ih, n = _case(rng)
        in_fmt = "yuvj420p" if rng.random() < 0.2 else "yuv420p"   # a full-range source: the same planes, other range steps
        if in_fmt == "yuvj420p" and rng.random() < 0.5 and "format" not in vf:
            vf += ",format=yuv420p"                            # (the reference's pin: the scale converts full -> limited inside its context)
        try:
            plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=in_fmt)
        except tts.TtsError:
            continue                                           # a chain the parser rejects (crop larger than the frame ...)
        seen.add(plan.kernel_name(0).split("<")[0])
        src = synth.frames("random", iw, ih, n)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d n=%d [%s | %s]: %d bytes differ, first at %d" % (
            vf, iw, ih, n, plan.kernel_name(0), plan.kernel_name(1), bad.size, bad[0])
        if rng.random() < 0.4:                                 # the same through per-plane pointers, tight strides
            ow, oh = plan.out_size
            keep, sp, ss, dp, ds = [], [], [], [], []
            for i in range(n):
                planes = [torch.from_numpy(np.ascontiguousarray(pl)).cuda() for pl in synth.split(src[i], iw, ih)]
                outs = [torch.full(((oh + k) // (1 + k) if k else oh, (ow + 1) // 2 if k else ow), 9, dtype=torch.uint8, device="cuda")
                        for k in (0, 1, 1)]
                keep += planes + outs
                sp.append([t.data_ptr() for t in planes]); ss.append([t.shape[1] for t in planes])
                dp.append(outs); ds.append([t.shape[1] for t in outs])
            plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
            torch.cuda.synchronize()
            for i in range(n):
                for k, w in enumerate(synth.split(want[i], ow, oh)):
                    assert np.array_equal(dp[i][k].cpu().numpy(), w), "pointer API: %s %dx%d n=%d frame %d plane %d [%s]" % (
                        vf, iw, ih, n, i, k, plan.kernel_name(0))
    assert seen, "no case ran"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_BIG_SEEDS", "6"))))
def test_random_geometries_larger_frames(tts, oracle, seed):
    """Frames up to ~1600 columns: long rows (two DMAs per row, paced workgroups, several row segments),
    padded strides through the pointer API."""
    import torch
    rng = np.random.default_rng(5000 + seed)
    for _ in range(5):
        iw, ih = int(rng.integers(300, 1700)), int(rng.integers(120, 700))
        r = float(rng.choice([rng.uniform(0.3, 1.0), rng.uniform(1.0, 4.0), rng.uniform(4.0, 14.0)]))
        ow, oh = max(4, int(iw / r)), max(4, int(ih / (r * rng.uniform(0.8, 1.25))))
        if ow * oh > 2_000_000:
            continue
        vf = "scale=%d:%d%s" % (ow, oh, rng.choice(["", ":flags=area", ""]))
        n = int(rng.choice([1, 3, 9]))
        plan = tts.Plan.from_vf(vf, iw, ih)
        src = synth.frames(str(rng.choice(["random", "impulse"])), iw, ih, n)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d n=%d [%s | %s]: %d bytes differ, first at %d" % (
            vf, iw, ih, n, plan.kernel_name(0), plan.kernel_name(1), bad.size, bad[0])
        # pointer API, padded strides
        keep, sp, ss, dp, ds = [], [], [], [], []
        pad_s, pad_d = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        for i in range(n):
            planes, strides = [], []
            for pl in synth.split(src[i], iw, ih):
                t = torch.zeros((pl.shape[0], pl.shape[1] + pad_s), dtype=torch.uint8, device="cuda")
                t[:, : pl.shape[1]] = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
                keep.append(t); planes.append(t.data_ptr()); strides.append(t.shape[1])
            outs = [torch.full(((oh + 1) // 2 if k else oh, ((ow + 1) // 2 if k else ow) + pad_d), 9, dtype=torch.uint8, device="cuda") for k in (0, 1, 1)]
            keep += outs
            sp.append(planes); ss.append(strides); dp.append(outs); ds.append([t.shape[1] for t in outs])
        plan.run_frames(sp, [[t.data_ptr() for t in o] for o in dp], ss, ds)
        torch.cuda.synchronize()
        for i in range(n):
            for k, w in enumerate(synth.split(want[i], ow, oh)):
                g = dp[i][k].cpu().numpy()
                assert np.array_equal(g[:, : w.shape[1]], w), "pointer API: %s %dx%d n=%d frame %d plane %d [%s]" % (vf, iw, ih, n, i, k, plan.kernel_name(0))
                assert (g[:, w.shape[1]:] == 9).all(), "wrote into the stride padding: %s" % vf


def _rgb_case(rng):
    """A chain that ends in packed RGB: mostly enlargements (the fused strip kernel: <= 4 taps an axis), some
    shrinks and conversions without a scale (generic / unscaled kernels), crops and pads on either side."""
    iw, ih = 2 * int(rng.integers(8, 260)), 2 * int(rng.integers(8, 200))
    kind = rng.choice(["up", "up", "up", "down", "none"])
    parts = []
    if kind != "none":
        if rng.random() < 0.25 and iw > 60 and ih > 60:
            cw, chh = int(rng.integers(24, iw)), int(rng.integers(24, ih))
            parts.append("crop=%d:%d:%d:%d" % (cw, chh, int(rng.integers(0, iw - cw + 1)), int(rng.integers(0, ih - chh + 1))))
        f = rng.uniform(1.0, 3.2) if kind == "up" else 1.0 / rng.uniform(1.0, 5.0)
        ow, oh = max(4, int(iw * f)), max(4, int(ih * f * rng.uniform(0.8, 1.25)))
        parts.append("scale=%d:%d%s" % (ow, oh, rng.choice(["", "", ":flags=bilinear"])))
        if rng.random() < 0.3 and ow > 24 and oh > 24:
            parts.append("crop=%d:%d" % (int(rng.integers(8, ow)), int(rng.integers(8, oh))))
        if rng.random() < 0.2:
            parts.append("pad=iw+%d:ih+%d:-1:-1:color=%s" % (int(rng.integers(0, 30)), int(rng.integers(0, 30)), rng.choice(["black", "red", "white"])))
    parts.append("format=%s" % rng.choice(["rgb24", "rgb24", "bgr24", "rgba", "bgra"]))
    return ",".join(parts), iw, ih, int(rng.choice([1, 2, 5, 33]))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_RGB_SEEDS", "12"))))
def test_random_geometries_packed_rgb_out(tts, oracle, seed):
    import torch
    rng = np.random.default_rng(9000 + seed)
    ran = 0
    for _ in range(12):
        vf, iw, ih, n = _rgb_case(rng)
        try:
            plan = tts.Plan.from_vf(vf, iw, ih)
        except tts.TtsError:
            continue                                           # crops that leave the frame, paths swscale takes that are not restated
        src = synth.frames(str(rng.choice(["random", "impulse", "zone"])), iw, ih, n)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d n=%d [%s]: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, n, plan.kernel_name(0), bad.size, bad[0], got.ravel()[bad[0]], want.ravel()[bad[0]])
        ran += 1
    assert ran, "no case ran"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_RGB_SEEDS", "12"))))
def test_random_small_shrinks_packed_rgb_out(tts, oracle, seed):
    """Shrinks between 1x and 2.2x an axis (5-8 taps) at sizes where the LDS strip kernels run: their 16-bit luma
    store and the in-place conversion behind it, with crops, pads and odd sizes; batches and one-off frames."""
    import torch
    rng = np.random.default_rng(12000 + seed)
    ran = 0
    for _ in range(6):
        iw, ih = 2 * int(rng.integers(40, 700)), 2 * int(rng.integers(30, 420))
        ow, oh = max(16, int(iw / rng.uniform(1.0, 2.2))), max(16, int(ih / rng.uniform(1.0, 2.2)))
        parts = ["scale=%d:%d%s" % (ow, oh, rng.choice(["", "", ":flags=bilinear", ":flags=area"]))]
        if rng.random() < 0.3:
            cw, chh = int(rng.integers(12, ow + 1)), int(rng.integers(12, oh + 1))
            parts.append("crop=%d:%d:%d:%d" % (cw, chh, int(rng.integers(0, ow - cw + 1)), int(rng.integers(0, oh - chh + 1))))
        if rng.random() < 0.3:
            parts.append("pad=iw+%d:ih+%d:-1:-1:color=%s" % (int(rng.integers(0, 40)), int(rng.integers(0, 40)), rng.choice(["black", "red", "white"])))
        parts.append("format=%s" % rng.choice(["rgb24", "rgb24", "bgr24", "rgba", "bgra"]))
        vf, n = ",".join(parts), int(rng.choice([1, 2, 3]))
        try:
            plan = tts.Plan.from_vf(vf, iw, ih)
        except tts.TtsError:
            continue
        src = synth.frames(str(rng.choice(["random", "impulse", "zone"])), iw, ih, n)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d n=%d [%s]: %d bytes differ, first at %d (got %d want %d)" % (
            vf, iw, ih, n, plan.kernel_name(0), bad.size, bad[0], got.ravel()[bad[0]], want.ravel()[bad[0]])
        ran += 1
    assert ran, "no case ran"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_RGB_SEEDS", "12"))))
def test_random_enlargements_packed_rgb_in(tts, oracle, seed):
    """rgb24 in, enlarged by 1.05x .. 4x an axis (independently), output widths on and off the strip walk's grid, crops
    inside the enlarged picture at odd offsets, bilinear now and then; batches of one and of many row segments."""
    import torch
    rng = np.random.default_rng(18000 + seed)
    ran = strip = 0
    for _ in range(6):
        iw, ih = 2 * int(rng.integers(8, 330)), 2 * int(rng.integers(6, 200))
        ow = int(iw * rng.uniform(1.05, 4.0))
        if rng.random() < 0.8:
            ow = max(8, ow // 8 * 8)
        oh = max(8, int(ih * rng.uniform(1.05, 4.0)))
        parts = ["scale=%d:%d%s" % (ow, oh, rng.choice(["", "", ":flags=bilinear"]))]
        if rng.random() < 0.35:
            cw = max(8, int(rng.integers(8, ow + 1)) // 8 * 8)
            chh = int(rng.integers(8, oh + 1))
            parts.append("crop=%d:%d:%d:%d" % (cw, chh, int(rng.integers(0, ow - cw + 1)), int(rng.integers(0, oh - chh + 1))))
        if rng.random() < 0.25:                                # what the frames leave as: full range out of the converting context, a pass of its own, or both
            parts.append(str(rng.choice(["format=yuvj420p", "format=yuv420p,format=yuvj420p", "format=yuvj420p,format=yuv420p"])))
        vf, n = ",".join(parts), int(rng.choice([1, 2, 5, 21]))
        try:
            plan = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
        except tts.TtsError:
            continue
        strip += plan.kernel_name(0).startswith("rgbin_strip")
        src = rng.integers(0, 256, (n, iw * ih * 3), dtype=np.uint8)
        m = min(n, 3)
        want = oracle.run_chain(plan.geom, src[:m])
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got[:m].ravel() != want.ravel())
        assert bad.size == 0, "%s %dx%d n=%d [%s]: %d bytes differ, first at %d" % (vf, iw, ih, n, plan.kernel_name(0), bad.size, bad[0])
        if n > m:                                              # the rest of a batch: frame k against frame k alone
            k = int(rng.integers(m, n))
            assert np.array_equal(got[k], plan.run_batch(torch.from_numpy(src[k: k + 1]).cuda()).cpu().numpy()[0])
        ran += 1
    assert ran, "no case ran"


def _repack(rgb, w, h, fmt):
    n = rgb.shape[0]
    px = rgb.reshape(n, h, w, 3)
    if fmt == "rgb24":
        return rgb
    if fmt == "bgr24":
        return np.ascontiguousarray(px[..., ::-1]).reshape(n, -1)
    out = np.empty((n, h, w, 4), np.uint8)
    out[..., :3] = px if fmt == "rgba" else px[..., ::-1]
    out[..., 3] = 0x5A
    return out.reshape(n, -1)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_RGB_SEEDS", "12"))))
def test_random_shrinks_packed_rgb_in(tts, oracle, seed):
    """Packed RGB in (all four layouts), shrunk by 1x .. 2.3x across and 1x .. 6x down (independently): the source-row walk
    where the geometry fits it (widths on and off its grid of 4 source pixels, output widths of 4 k and 4 k + 2, crops
    behind the scale at odd rows), the tiled kernel elsewhere; batches whose waves straddle frames."""
    import torch
    rng = np.random.default_rng(21000 + seed)
    ran = walk = 0
    for _ in range(6):
        iw = int(rng.choice([4, 4, 2])) * int(rng.integers(8, 360))
        ih = int(rng.integers(12, 420))
        ow = max(8, int(iw / (rng.uniform(1.0, 2.3) if rng.random() < 0.6 else rng.uniform(2.3, 12.5))))   # (beyond 8 taps across: two columns a lane)
        if rng.random() < 0.85:
            ow &= ~1
        oh = max(4, int(ih / rng.uniform(1.0, 6.0)))
        parts = ["scale=%d:%d%s" % (ow, oh, rng.choice(["", "", ":flags=bilinear", ":flags=area"]))]
        if rng.random() < 0.3 and ow > 40 and oh > 16:
            cw, chh = 2 * int(rng.integers(8, ow // 2 + 1)), int(rng.integers(8, oh + 1))
            parts.append("crop=%d:%d:%d:%d" % (cw, chh, 2 * int(rng.integers(0, (ow - cw) // 2 + 1)), int(rng.integers(0, oh - chh + 1))))
        if rng.random() < 0.2:                                 # a pad behind the scale: the border-fill launch in front of the walk
            cw_, ch_ = (int(parts[-1].split("=")[1].split(":")[0]), int(parts[-1].split("=")[1].split(":")[1])) if len(parts) > 1 else (ow, oh)
            pw, ph = cw_ + 2 * int(rng.integers(0, 20)), ch_ + 2 * int(rng.integers(0, 12))
            parts.append("pad=%d:%d:%s:color=%s" % (pw, ph, rng.choice(["-1:-1", "%d:%d" % (int(rng.integers(0, pw - cw_ + 1)), int(rng.integers(0, ph - ch_ + 1)))]), rng.choice(["black", "red", "0x336699"])))
        fmt = str(rng.choice(["rgb24", "rgb24", "bgr24", "rgba", "bgra"]))
        if rng.random() < 0.25:
            parts.append(str(rng.choice(["format=yuvj420p", "format=yuv420p,format=yuvj420p", "format=yuvj420p,format=yuv420p"])))
        vf, n = ",".join(parts), int(rng.choice([1, 2, 5, 19]))
        try:
            plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=fmt)
        except tts.TtsError:
            continue
        walk += plan.kernel_name(0).startswith("rgbin_shrink")
        src = _repack(rng.integers(0, 256, (n, iw * ih * 3), dtype=np.uint8), iw, ih, fmt)
        m = min(n, 3)
        want = oracle.run_chain(plan.geom, src[:m])
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got[:m].ravel() != want.ravel())
        assert bad.size == 0, "%s %dx%d %s n=%d [%s]: %d bytes differ, first at %d" % (vf, iw, ih, fmt, n, plan.kernel_name(0), bad.size, bad[0])
        if n > m:                                              # the rest of a batch: frame k against frame k alone
            k = int(rng.integers(m, n))
            assert np.array_equal(got[k], plan.run_batch(torch.from_numpy(src[k: k + 1]).cuda()).cpu().numpy()[0])
        ran += 1
    assert ran, "no case ran"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_RGB_SEEDS", "12"))))
def test_random_conversion_alone_both_directions(tts, oracle, seed):
    """No scale in the chain: yuv420p -> packed RGB and rgb24 -> yuv420p at random sizes -- widths on and off the grids
    the streaming kernels need, crops in front, short and long batches (one and many row segments)."""
    import torch
    rng = np.random.default_rng(15000 + seed)
    ran = 0
    for _ in range(6):
        iw = int(rng.choice([16, 8, 2])) * int(rng.integers(2, 90))
        iw = max(16, min(iw, 1440))
        ih = 2 * int(rng.integers(4, 300))
        parts = []
        if rng.random() < 0.3 and iw > 48 and ih > 48:
            cw, chh = 2 * int(rng.integers(8, iw // 2)), 2 * int(rng.integers(8, ih // 2))
            parts.append("crop=%d:%d:%d:%d" % (cw, chh, int(rng.choice([8, 2, 1])) * int(rng.integers(0, (iw - cw) // 8 + 1)), int(rng.integers(0, ih - chh + 1))))
        n = int(rng.choice([1, 2, 9, 33]))
        for rgb_in in (False, True):
            vf = ",".join(parts + ["format=" + ("yuv420p" if rgb_in else str(rng.choice(["rgb24", "bgr24", "rgba", "bgra"])))])
            in_fmt = str(rng.choice(["rgb24", "bgr24"])) if rgb_in else "yuv420p"   # (bgr24 at the same size: swscale's own converter for that pair, bgr24_yv12*_kernel)
            if rgb_in and rng.random() < 0.3:                 # a pad on the RGB pixels in front of the conversion pass (any offset, even canvas width)
                vf = ",".join(parts + ["pad=iw+%d:ih+%d:%d:%d:color=%s" % (2 * int(rng.integers(1, 9)) + (cw & 1 if parts else 0), int(rng.integers(1, 17)), int(rng.integers(0, 3)), int(rng.integers(0, 2)), rng.choice(["black", "red", "0x336699"]))])
            try:
                plan = tts.Plan.from_vf(vf, iw, ih, in_fmt=in_fmt)
            except tts.TtsError:
                continue
            if rgb_in:
                src = rng.integers(0, 256, (n, iw * ih * 3), dtype=np.uint8)
            else:
                src = synth.frames(str(rng.choice(["random", "impulse"])), iw, ih, n)
            want = oracle.run_chain(plan.geom, src)
            got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
            bad = np.flatnonzero(got.ravel() != want.ravel())
            assert bad.size == 0, "%s %dx%d n=%d rgb_in=%s [%s]: %d bytes differ, first at %d" % (vf, iw, ih, n, rgb_in, plan.kernel_name(0), bad.size, bad[0])
            ran += 1
    assert ran, "no case ran"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TTS_FUZZ_RGB_SEEDS", "12"))))
def test_random_geometries_packed_rgb_in(tts, oracle, seed):
    import torch
    rng = np.random.default_rng(11000 + seed)
    ran = 0
    for _ in range(10):
        iw, ih = 2 * int(rng.integers(8, 200)), int(rng.integers(12, 300))
        f = float(rng.choice([rng.uniform(1.0, 3.0), 1.0 / rng.uniform(1.0, 8.0)]))
        ow, oh = max(4, int(iw * f)), max(4, int(ih * f * rng.uniform(0.8, 1.25)))
        vf = "scale=%d:%d%s" % (ow, oh, rng.choice(["", ":flags=area", ""]))
        if rng.random() < 0.3 and ow > 24 and oh > 24:
            vf += ",crop=%d:%d" % (int(rng.integers(8, ow)), int(rng.integers(8, oh)))
        if rng.random() < 0.25:
            vf += ",pad=iw+%d:ih+%d:-1:-1:color=black" % (2 * int(rng.integers(0, 12)), 2 * int(rng.integers(0, 12)))
        n = int(rng.choice([1, 3, 17]))
        try:
            plan = tts.Plan.from_vf(vf, iw, ih, in_fmt="rgb24")
        except tts.TtsError:
            continue
        src = np.random.default_rng(seed).integers(0, 256, (n, iw * ih * 3), dtype=np.uint8)
        want = oracle.run_chain(plan.geom, src)
        got = plan.run_batch(torch.from_numpy(src).cuda()).cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s %dx%d n=%d [%s]: %d bytes differ, first at %d" % (vf, iw, ih, n, plan.kernel_name(0), bad.size, bad[0])
        ran += 1
    assert ran, "no case ran"
