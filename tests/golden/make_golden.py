#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/swscale_oracle.c).

The reference (forestrebel/tiktoken) holds no pixel vectors and FFmpeg is not available
(SURVEY.md 8c), so these are regression vectors of the project's own oracle: PARITY UNPINNED
against real swscale.  Inputs are the seeded synthetic frames of tests/synth.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from oracle import oracle  # noqa: E402

# (name, geometry dict) -- geometry spelled out so that this file does not depend on the product's parser
def geom(in_w, in_h, sc_w, sc_h, alg=0, pre=None, crop=None, pad=None, fill=(16, 128, 128), out_range=0, rgb=None):
    pre = pre or (0, 0, in_w, in_h)
    crop = crop or (0, 0, sc_w, sc_h)
    g = dict(in_w=in_w, in_h=in_h, pre_x=pre[0], pre_y=pre[1], pre_w=pre[2], pre_h=pre[3],
             do_scale=int((sc_w, sc_h) != (pre[2], pre[3])), alg=alg, sc_w=sc_w, sc_h=sc_h,
             crop_x=crop[0], crop_y=crop[1], crop_w=crop[2], crop_h=crop[3],
             do_pad=0, pad_x=0, pad_y=0, out_w=crop[2], out_h=crop[3],
             fill_y=fill[0], fill_u=fill[1], fill_v=fill[2])
    if out_range:
        g["out_range"] = out_range                     # (left out when 0 so the older fixtures keep their key lists)
    if pad:
        g.update(do_pad=1, pad_x=pad[0], pad_y=pad[1], out_w=pad[2], out_h=pad[3])
    if rgb:                                            # (out_fmt, rgb_stage, fill_rgb): packed RGB written by the chain
        g.update(out_fmt=rgb[0], rgb_stage=rgb[1], fill_rgb=rgb[2])
    return g


CASES = {
    # scaled-down relatives of the reference's and BASELINE.json's chains (same ratios, same phases)
    "portrait_192x108": geom(192, 108, 341, 192, crop=(116, 0, 108, 192)),          # 1080p->portrait / 10
    "portrait_from_4k_384x216": geom(384, 216, 341, 192, crop=(116, 0, 108, 192)),
    "stretch_96x170_to_144x256": geom(96, 170, 144, 256),                            # processVideos.js:96 / 5
    "thumb_pad_72x128": geom(72, 128, 101, 180, pad=(108, 0, 320, 180)),             # video.js:329-332, small source
    "thumb_area_108x192": geom(108, 192, 32, 57, alg=1),                             # 320x568 area / 10
    "bilinear_half": geom(128, 72, 64, 36, alg=2),
    "precrop_scale": geom(100, 60, 128, 96, pre=(10, 6, 64, 32)),
    "identity": geom(64, 48, 64, 48),
    "thumb_pad_jpeg_72x128": geom(72, 128, 101, 180, pad=(108, 0, 320, 180), out_range=1),   # ... then the mjpeg full-range pass
    # packed RGB: the scale writes rgb24 (pad on RGB pixels, picture at the odd x = 109), and the conversion-only pass
    "portrait_rgb24_192x108": geom(192, 108, 341, 192, crop=(116, 0, 108, 192), rgb=(1, 1, 0)),
    "thumb_pad_rgb24_72x128": geom(72, 128, 101, 180, pad=(109, 0, 320, 180), rgb=(1, 1, 0)),
    "convert_only_bgra_64x48": geom(64, 48, 64, 48, rgb=(4, 2, 0)),
    # round 3: the scale itself writes yuvj420p (range step on its 15-bit rows, between the passes; the pad colour in full range)
    "scale_writes_yuvj_72x128": dict(geom(72, 128, 101, 180, pad=(108, 0, 320, 180), fill=(0, 128, 128)), sc_range=1),
    # round 3: a full-range source through the thumbnail chain: full -> limited inside the scale (sc_range 2), back to full at the end (out_range 1)
    "thumb_pad_full_range_source_72x128": dict(geom(72, 128, 101, 180, pad=(108, 0, 320, 180), out_range=1), sc_range=2),
    # round 3: rows through swscale's 2-line writer (bilinear enlargement to RGB), and RGB from a full-range link
    "bilinear_rgb24_64x48_to_128x96": geom(64, 48, 128, 96, alg=2, rgb=(1, 1, 0)),
    "thumb_pad_jpeg_rgb24_72x128": geom(72, 128, 101, 180, pad=(108, 0, 320, 180), out_range=1, rgb=(1, 2, 0)),
}
# packed rgb24 IN (yuv420p out): inputs are seeded random bytes, 3 a pixel
RGBIN_CASES = {
    "rgbin_stretch_96x170_to_144x256": dict(geom(96, 170, 144, 256), in_fmt=1),
    "rgbin_convert_only_64x48": dict(geom(64, 48, 64, 48), in_fmt=1),
    "rgbin_portrait_192x108": dict(geom(192, 108, 341, 192, crop=(116, 0, 108, 192)), in_fmt=1),   # widened: chroma read per pixel (rgb24ToUV_c)
    "rgbin_shrink_96x170_to_54x96": dict(geom(96, 170, 54, 96), in_fmt=1),                          # shrunk: chroma read from pixel pairs
    "rgbin_pad_on_rgb_64x36_in_80x61": dict(geom(64, 36, 64, 36, pad=(7, 5, 80, 61)), in_fmt=1, rgb_stage=2, fill_rgb=0x336699, out_fmt=0),   # no scale: the pad on RGB pixels, then the conversion of the canvas
    "rgbin_to_yuvj_96x170_to_54x96": dict(geom(96, 170, 54, 96), in_fmt=1, sc_range=1),            # the converting context writes full range (range step on its 15-bit rows)
    "rgbin_thumb_pad_then_range_pass_72x128": dict(geom(72, 128, 101, 180, pad=(108, 0, 320, 180), out_range=1), in_fmt=1),   # the reference's thumbnail chain + its encoder's range pass, fed with rgb24
    "bgr24_same_size_66x37": dict(geom(66, 37, 66, 37), in_fmt=2),                                 # swscale's bgr24toYV12 converter (ff_rgb24toyv12_c), odd height
}
PATTERNS = ("random", "impulse", "zone")


def main():
    for name, g in CASES.items():
        arrs = {"geom": np.array([g[k] for k in sorted(g)], np.int32), "geom_keys": np.array(sorted(g))}
        for pat in PATTERNS:
            src = synth.frames(pat, g["in_w"], g["in_h"], 1)
            arrs["out_" + pat] = oracle.run_chain(g, src)[0]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
        print(name, {k: v.shape for k, v in arrs.items() if k.startswith("out_")})
    for name, g in RGBIN_CASES.items():
        src = np.random.default_rng(synth.SEED).integers(0, 256, (1, g["in_w"] * g["in_h"] * 3), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), geom=np.array([g[k] for k in sorted(g)], np.int32),
                            geom_keys=np.array(sorted(g)), out_rgbseed=oracle.run_chain(g, src)[0])
        print(name)
    # coefficient tables of the headline geometries (first/last rows are where the border rules bite)
    tabs = {}
    for (s, d, one, alg) in [(1920, 3413, 1 << 14, 0), (1080, 1920, 1 << 12, 0), (960, 1707, 1 << 14, 0),
                             (3840, 3413, 1 << 14, 0), (1080, 320, 1 << 14, 1), (1920, 568, 1 << 12, 1),
                             (720, 101, 1 << 14, 0), (1280, 180, 1 << 12, 0)]:
        c, p = oracle.filter_table(s, d, one, alg)
        tabs["c_%d_%d_%d_%d" % (s, d, one, alg)] = c
        tabs["p_%d_%d_%d_%d" % (s, d, one, alg)] = p
    np.savez_compressed(os.path.join(HERE, "filter_tables.npz"), **tabs)


if __name__ == "__main__":
    main()
