"""The oracle against the textbook definitions, in float64: a separable centre-aligned Keys cubic
(a = -0.6, kernel widened by the ratio when shrinking, taps that fall off the image folded onto the
edge sample) and the box filter of `area`.  This does not pin the oracle to libswscale bit for bit
(nothing here can: DESIGN.md 2) -- it pins what the fixed-point tables and shifts are an approximation
OF: a half-pixel shift, a wrong cubic parameter or a transposed axis shows up as tens of levels on
noise; coefficient quantisation (14 / 12 bits), the truncating >>7 and tap trimming as a level or two."""
import numpy as np
import pytest

import synth


def keys(x, a=-0.6):
    x = np.abs(x)
    return np.where(x <= 1, (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1,
                    np.where(x < 2, a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a, 0.0))


def box(x):
    return np.where(np.abs(x) <= 0.5, 1.0, 0.0)


def axis_matrix(n_src, n_dst, kind):
    """[n_dst, n_src] float weights of one axis."""
    ratio = n_src / n_dst
    m = np.zeros((n_dst, n_src))
    j = np.arange(-8 * int(np.ceil(max(ratio, 1))) - 8, n_src + 8 * int(np.ceil(max(ratio, 1))) + 8)
    for i in range(n_dst):
        centre = (i + 0.5) * ratio - 0.5
        if kind == "bicubic":
            w = keys((j - centre) / max(ratio, 1.0))
        else:                                             # area: overlap of source sample j's cell with the output cell
            lo, hi = i * ratio, (i + 1) * ratio
            w = np.clip(np.minimum(j + 1, hi) - np.maximum(j, lo), 0, None) if ratio >= 1 else None
            if w is None:                                 # (enlarging with `area` is bilinear in swscale: not compared here)
                raise ValueError
        w = w / w.sum()
        np.add.at(m[i], np.clip(j, 0, n_src - 1), w)      # fold taps beyond the edges onto the edge sample
    return m


def float_plane(p, ow, oh, kind):
    h = axis_matrix(p.shape[1], ow, kind)
    v = axis_matrix(p.shape[0], oh, kind)
    mid = np.minimum(p.astype(np.float64) @ h.T, 32767 / 128)      # swscale's 15-bit intermediate: clipped above, not below
    return np.clip(v @ mid, 0, 255)


@pytest.mark.parametrize("vf,iw,ih,kind,tol", [
    ("scale=320:180", 96, 54, "bicubic", 2.0),                  # 3.3x up
    ("scale=180:320:force_original_aspect_ratio=increase,crop=180:320", 320, 180, "bicubic", 2.0),
    ("scale=101:57", 64, 36, "bicubic", 2.0),                   # odd sizes
    ("scale=150:84", 200, 112, "bicubic", 2.5),                 # 1.33x down
    ("scale=96:54", 320, 180, "bicubic", 3.0),                  # 3.3x down: 15 taps, trimmed
    ("scale=80:142:flags=area", 270, 480, "area", 2.0),         # the thumbnail ratio
    ("scale=64:36:flags=area", 256, 144, "area", 1.5),          # 4x: exact box
])
def test_oracle_is_the_textbook_resampler_within_rounding(tts, oracle, vf, iw, ih, kind, tol):
    g = tts.parse_vf(vf, iw, ih)
    src = synth.frames("random", iw, ih, 1)
    got = oracle.run_chain(g, src)[0]
    gd = g.as_dict()
    worst = 0.0
    for k, (pl, gpl) in enumerate(zip(synth.split(src[0], iw, ih), synth.split(got, gd["out_w"], gd["out_h"]))):
        sub = 0 if k == 0 else 1
        sw, sh = (gd["sc_w"] + sub) >> sub, (gd["sc_h"] + sub) >> sub
        ref = float_plane(pl, sw, sh, kind)
        cx, cy = gd["crop_x"] >> sub, gd["crop_y"] >> sub
        ref = ref[cy: cy + gpl.shape[0], cx: cx + gpl.shape[1]]
        diff = np.abs(ref - gpl.astype(np.float64))
        # The first outputs of an enlarged axis are left out: swscale places the first tap with a division
        # that truncates toward zero (initFilter: `xx = (xDstInSrc - ...) / (1 << 17)`), so where that
        # position is negative the window starts one sample late and is renormalised -- the oracle restates
        # that (measured here: row 3 of 96 -> 320 carries 0.475 / 0.601 / -0.077 instead of the textbook
        # 0.440 / 0.642 / -0.082); the textbook does not.
        mx = int(np.ceil(1.5 * sw / pl.shape[1])) if sw > pl.shape[1] else 0
        my = int(np.ceil(1.5 * sh / pl.shape[0])) if sh > pl.shape[0] else 0
        if cx == 0: diff[:, :mx] = 0
        if cy == 0: diff[:my, :] = 0
        worst = max(worst, float(diff.max()))
        assert diff.mean() < 0.6, (vf, k, diff.mean())
    assert worst <= tol, (vf, worst)


def test_range_step_is_the_textbook_mapping_within_one_level(tts, oracle):
    """format=yuvj420p after the chain: limited -> full range.  Textbook: Y' = (Y - 16) * 255 / 219,
    C' = (C - 128) * 255 / 224 + 128, clipped.  swscale's fixed-point constants (recalled) may round
    differently by a level, and clip the input to the nominal range first."""
    g = tts.parse_vf("format=yuvj420p", 16, 16)
    frames = np.zeros((4, 16 * 16 + 2 * 64), np.uint8)
    for f in range(4):
        frames[f, :256] = np.arange(256)
        frames[f, 256:320] = np.arange(64) + 64 * f
        frames[f, 320:] = np.arange(64) + 64 * f
    got = oracle.run_chain(g, frames)
    y = got[0, :256].astype(np.float64)
    c = np.concatenate([got[f, 256:320] for f in range(4)]).astype(np.float64)
    p = np.arange(256, dtype=np.float64)
    assert np.abs(y - np.clip((p - 16) * 255 / 219, 0, 255)).max() <= 1.0
    assert np.abs(c - np.clip((np.clip(p, 16, 240) - 128) * 255 / 224 + 128, 0, 255)).max() <= 1.0
    assert np.array_equal(np.concatenate([got[f, 320:] for f in range(4)]), c.astype(np.uint8))


def test_rgb_tables_are_bt601_within_a_stated_bound(tts, oracle):
    """format=rgb24 alone: swscale's unscaled converter over its yuv2rgb tables, as recalled (ITU-R BT.601,
    limited range; oracle/swscale_oracle.c tto_yuv2rgb_init).  Textbook: R = 1.164 (Y - 16) + 1.596 (V - 128),
    G = 1.164 (Y - 16) - 0.392 (U - 128) - 0.813 (V - 128), B = 1.164 (Y - 16) + 2.017 (U - 128), clipped.
    STATED BOUND: within 3 levels, and on average between 0 and 2 levels DARKER than the textbook: the recalled
    table origin (yoffs = 326 against a 384 << 16 ramp start) sits 1.5 luma steps low, and the chroma offsets
    are floored twice.  A wrong matrix, swapped U / V or a full-range table shows up as tens of levels."""
    w, h = 64, 48
    g = tts.parse_vf("format=rgb24", w, h)
    src = synth.frames("limited", w, h, 4)
    got = oracle.run_chain(g, src).reshape(4, h, w, 3).astype(np.float64)
    for i in range(4):
        y, u, v = (p.astype(np.float64) for p in synth.split(src[i], w, h))
        u = np.repeat(np.repeat(u, 2, 0), 2, 1) - 128
        v = np.repeat(np.repeat(v, 2, 0), 2, 1) - 128
        yy = 255.0 / 219.0 * (y - 16)
        ref = np.clip(np.stack([yy + 1.596027 * v, yy - 0.391762 * u - 0.812968 * v, yy + 2.017232 * u], -1), 0, 255)
        d = got[i] - ref
        inside = (ref > 3) & (ref < 252)                      # away from the clip points, where a shift is a shift
        assert np.abs(d).max() <= 3.0, np.abs(d).max()
        assert -2.0 <= d[inside].mean() <= 0.0, d[inside].mean()
    # grey ramp: neutral chroma gives R = G = B, monotone in Y, black at 16 and (recalled tables) 253 at 235
    ramp = np.zeros((1, w * h + 2 * (w // 2) * (h // 2)), np.uint8)
    ramp[0, : w * h] = np.arange(w * h) % 256
    ramp[0, w * h:] = 128
    px = oracle.run_chain(g, ramp).reshape(h * w, 3)[:256]
    assert (px[:, 0] == px[:, 1]).all() and (px[:, 1] == px[:, 2]).all()
    assert (np.diff(px[:, 0].astype(int)) >= 0).all() and px[16, 0] == 0 and px[235, 0] in (253, 254, 255) and px[255, 0] == 255


def test_rgb_scaled_path_agrees_with_scaling_then_converting_in_float(tts, oracle):
    """scale=...,format=rgb24: luma to W x H, chroma to ceil(W / 2) x H (two pixels share a sample, every row its
    own), then the matrix -- against the float resampler above followed by the float matrix.  STATED BOUND: 8.5
    levels worst case, 1.6 on average: the planes themselves agree with the float resampler to 2 levels (first
    test of this file), the matrix multiplies a chroma error by up to 2.017 (B) and a luma error by 1.164, and the
    recalled tables sit 1.8 levels dark: 2 x 2.017 + 2 x 1.164 + 1.8."""
    iw, ih, ow, oh = 96, 54, 200, 112
    g = tts.parse_vf("scale=%d:%d,format=rgb24" % (ow, oh), iw, ih)
    src = synth.frames("limited", iw, ih, 1)
    got = oracle.run_chain(g, src).reshape(oh, ow, 3).astype(np.float64)
    y, u, v = synth.split(src[0], iw, ih)
    yf = float_plane(y, ow, oh, "bicubic")
    uf = np.repeat(float_plane(u, (ow + 1) // 2, oh, "bicubic"), 2, 1)[:, :ow] - 128
    vf_ = np.repeat(float_plane(v, (ow + 1) // 2, oh, "bicubic"), 2, 1)[:, :ow] - 128
    yy = 255.0 / 219.0 * (yf - 16)
    ref = np.clip(np.stack([yy + 1.596027 * vf_, yy - 0.391762 * uf - 0.812968 * vf_, yy + 2.017232 * uf], -1), 0, 255)
    d = np.abs(got - ref)[8:, 8:]                              # (first outputs of an enlarged axis: see above)
    assert d.max() <= 8.5 and d.mean() < 1.6, (d.max(), d.mean())


def test_rgb24_input_is_bt601_within_a_level(tts, oracle):
    """rgb24 in, yuv420p out, no scaling: Y = 16 + 219/255 (0.299 R + 0.587 G + 0.114 B) to a level; chroma = the
    BT.601 difference signals of the average of two neighbouring pixels, resampled 2:1 down the rows by the cubic
    (every RGB row carries chroma), within 2 levels of the float resampler."""
    w, h = 64, 48
    g = tts.parse_vf("format=yuv420p", w, h, in_fmt="rgb24")
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    y, u, v = synth.split(oracle.run_chain(g, rgb.reshape(1, -1))[0], w, h)
    r, gg, b = (rgb[0, :, :, k].astype(np.float64) for k in range(3))
    assert np.abs(y - (16 + 219.0 / 255 * (0.299 * r + 0.587 * gg + 0.114 * b))).max() <= 1.0
    pair = lambda p: (p[:, 0::2] + p[:, 1::2]) / 2
    cb = 128 + 224.0 / 255 * (-0.169 * pair(r) - 0.331 * pair(gg) + 0.5 * pair(b))
    cr = 128 + 224.0 / 255 * (0.5 * pair(r) - 0.419 * pair(gg) - 0.081 * pair(b))
    vm = axis_matrix(h, h // 2, "bicubic")
    assert np.abs(u - np.clip(vm @ cb, 0, 255)).max() <= 2.0 and np.abs(v - np.clip(vm @ cr, 0, 255)).max() <= 2.0


def test_rgb24_input_widened_reads_chroma_per_pixel(tts, oracle):
    """rgb24 in, widened (480x854 -> 720x1280 in miniature): sws_init_context() keeps chrSrcHSubSample = 0 when the
    chroma output is wider than half the source, so chroma is one BT.601 difference sample PER PIXEL (rgb24ToUV_c),
    shrunk src_w -> ceil(dst_w / 2) across and src_h -> ceil(dst_h / 2) down by the cubic -- against the float
    resampler, within 2 levels; the pair-averaging reading (round 2's oracle) differs from it by more than that on
    noise, which the last assertion shows so that the test can tell the two apart."""
    iw, ih, ow, oh = 48, 40, 72, 60
    g = tts.parse_vf("scale=%d:%d" % (ow, oh), iw, ih, in_fmt="rgb24")
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (1, ih, iw, 3), dtype=np.uint8)
    y, u, v = synth.split(oracle.run_chain(g, rgb.reshape(1, -1))[0], ow, oh)
    r, gg, b = (rgb[0, :, :, k].astype(np.float64) for k in range(3))
    cb = 128 + 224.0 / 255 * (-0.169 * r - 0.331 * gg + 0.5 * b)
    cr = 128 + 224.0 / 255 * (0.5 * r - 0.419 * gg - 0.081 * b)
    hm, vm = axis_matrix(iw, ow // 2, "bicubic"), axis_matrix(ih, oh // 2, "bicubic")
    fu, fv = np.clip(vm @ (cb @ hm.T), 0, 255), np.clip(vm @ (cr @ hm.T), 0, 255)
    assert np.abs(u - fu).max() <= 2.0 and np.abs(v - fv).max() <= 2.0, (np.abs(u - fu).max(), np.abs(v - fv).max())
    pair = lambda p: (p[:, 0::2] + p[:, 1::2]) / 2
    hm2 = axis_matrix(iw // 2, ow // 2, "bicubic")
    assert np.abs(u - np.clip(vm @ (pair(cb) @ hm2.T), 0, 255)).max() > 4.0
    lum = 16 + 219.0 / 255 * (0.299 * r + 0.587 * gg + 0.114 * b)
    fy = np.clip(axis_matrix(ih, oh, "bicubic") @ (lum @ axis_matrix(iw, ow, "bicubic").T), 0, 255)
    assert np.abs(y - fy)[4:, 4:].max() <= 2.0


def test_rgb24_input_chroma_rule(oracle):
    """(dst_w >> 1) <= (src_w >> 1) picks the pixel-pair reader: the conversion alone, every shrink, and a widening
    by one column of an even width; everything wider reads chroma per pixel."""
    L = oracle.lib()
    assert [L.tto_rgb_in_half_chroma(sw, dw) for sw, dw in ((64, 64), (64, 32), (64, 65), (64, 66), (1920, 3413), (480, 720), (63, 64), (63, 63))] == [1, 1, 1, 0, 0, 0, 0, 1]


def test_rgb_from_full_range_is_the_jfif_matrix(tts, oracle):
    """format=yuvj420p,format=rgb24: the conversion pass reads a FULL-range source, so swscale builds its tables with
    fullRange = 1 -- no luma stretch, chroma coefficients x 224 / 255: R = Y + 1.402 (V - 128),
    G = Y - 0.344 (U - 128) - 0.714 (V - 128), B = Y + 1.772 (U - 128), the JFIF matrix -- applied to the planes the
    range pass leaves (the oracle's own `format=yuvj420p` output).  STATED BOUND: 2 levels (two floored chroma
    products, two floored table offsets and one rounding: 1.54 measured); a limited-range table on this source would be off by up to 16 + 14 % of the level."""
    w, h = 64, 48
    src = synth.frames("random", w, h, 2)
    rgb = oracle.run_chain(tts.parse_vf("format=yuvj420p,format=rgb24", w, h), src).reshape(2, h, w, 3).astype(np.float64)
    yuvj = oracle.run_chain(tts.parse_vf("format=yuvj420p", w, h), src)
    for i in range(2):
        y, u, v = (p.astype(np.float64) for p in synth.split(yuvj[i], w, h))
        u = np.repeat(np.repeat(u, 2, 0), 2, 1) - 128
        v = np.repeat(np.repeat(v, 2, 0), 2, 1) - 128
        ref = np.clip(np.stack([y + 1.402 * v, y - 0.344136 * u - 0.714136 * v, y + 1.772 * u], -1), 0, 255)
        assert np.abs(rgb[i] - ref).max() <= 2.0, np.abs(rgb[i] - ref).max()
