#!/usr/bin/env python3
"""Tuning sweep for the strip kernel: env knobs x geometries, one process, interleaved rounds.
Usage: python tools/sweep.py [--frames 300] [--rounds 5] [--case portrait|thumb|4k|720p]"""
import argparse
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiktoken_amd as tts

CASES = {
    "portrait": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1920, 1080, 4095360),
    "4k": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 3840, 2160, 7050240),
    "720p": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1280, 720, 3548880),
    "w1088": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1088:1920", 1920, 1080, 613 * 1080 * 3 // 2 + 1088 * 1920 * 3 // 2),
    "w1152": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1152:1920", 1920, 1080, 649 * 1080 * 3 // 2 + 1152 * 1920 * 3 // 2),
    "w512": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=512:1920", 1920, 1080, 291 * 1080 * 3 // 2 + 512 * 1920 * 3 // 2),     # 64 strips a row: a luma wave is one whole row, row after row
    "aligned": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1024:1920", 1920, 1080, 3882240),
    # the reference's own thumbnail chain (src/services/video.js:329-332) on a 1080p / portrait 1080p / 4K source
    "thumbref": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1920, 1080, 3110400 + 86400),
    "thumbref_portrait": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1080, 1920, 3110400 + 86400),
    "thumbref_portrait_nopad": ("scale=320x180:force_original_aspect_ratio=decrease", 1080, 1920, 3110400 + 101 * 180 + 2 * 51 * 90),
    "thumbref_4k": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 3840, 2160, 12441600 + 86400),
    # web compress (portrait.test.js:70-97): a 1080x1920 phone clip into the 720x1280 box; 2160x3840 likewise
    "web": ("scale=720:1280", 1080, 1920, 3110400 + 1382400),
    "web_4k": ("scale=720:1280", 2160, 3840, 12441600 + 1382400),
    "half": ("scale=1920:1080", 3840, 2160, 12441600 + 3110400),        # 4K -> 1080p, 2x: 9 taps
    "third": ("scale=1280:720", 3840, 2160, 12441600 + 1382400),        # 4K -> 720p, 3x: 13 taps
    "stretch": ("scale=w=720:h=1280", 480, 854, 614880 + 1382400),
    "thumb": ("scale=320:568:flags=area", 1080, 1920, 3383040),
    # area thumbnails at larger ratios: 6x (1080p -> 320x180), 4x (720p), 12x (4K)
    "area6": ("scale=320:180:flags=area", 1920, 1080, 3110400 + 86400),
    "area4": ("scale=320:180:flags=area", 1280, 720, 1382400 + 86400),
    "bil4": ("scale=320:180:flags=bilinear", 1280, 720, 1382400 + 86400),
    "area12": ("scale=320:180:flags=area", 3840, 2160, 12441600 + 86400),
    # packed RGB out: the same source window read once + 3 (4) bytes a pixel written once
    "portrait_rgb24": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=rgb24", 1920, 1080, 984960 + 6220800),
    "portrait_rgba": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=rgba", 1920, 1080, 984960 + 8294400),
    "720p_rgb24": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=rgb24", 1280, 720, 438480 + 6220800),
    "stretch_rgb24": ("scale=w=720:h=1280,format=rgb24", 480, 854, 614880 + 2764800),
    # packed rgb24 IN, yuv420p out (the reverse of the above): 3 bytes a pixel read once + the planes written once
    "portrait_rgbin": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920", 1920, 1080, 608 * 1080 * 3 + 3110400),
    "stretch_rgbin": ("scale=w=720:h=1280", 480, 854, 480 * 854 * 3 + 1382400),
    "half_rgbin": ("scale=960:540", 1920, 1080, 6220800 + 777600),                       # 2x shrink from RGB: 8 taps across, 16 down the chroma planes
    "web_rgbin": ("scale=720:1280", 1080, 1920, 6220800 + 1382400),                     # the web box from RGB: 1.5x
    "thumb_rgbin": ("scale=320:568:flags=area", 1080, 1920, 6220800 + 272640),          # BASELINE's thumbnails from RGB
    "thumbref_rgbin": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1920, 1080, 6220800 + 86400),   # the reference's thumbnail chain fed with rgb24 frames
    "thumbref_rgbin_yuvj": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p,format=yuvj420p", 1920, 1080, 6220800 + 86400),   # ... + the mjpeg encoder's range pass
    "thumbref_rgbin_inctx": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuvj420p", 1920, 1080, 6220800 + 86400),   # ... unpinned: the converting context writes full range itself
    "thumbref_rgbain": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1920, 1080, 8294400 + 86400),   # ... from rgba frames
    "thumbref_portrait_rgbin": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 1080, 1920, 6220800 + 86400),
    "portrait_rgbin_inctx": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=yuvj420p", 1920, 1080, 608 * 1080 * 3 + 3110400),
    "quarter_rgbin": ("scale=480:270", 1920, 1080, 6220800 + 194400),                   # 4x from RGB: 17 taps
    "portrait_rgbin_yuvj": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=yuv420p,format=yuvj420p", 1920, 1080, 608 * 1080 * 3 + 3110400),
    "cvt_yuvj": ("format=yuvj420p", 1920, 1080, 3110400 * 2),                               # range step alone
    "thumbref_yuvj": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p,format=yuvj420p", 1920, 1080, 3110400 + 86400),   # the reference's chain + the mjpeg encoder's range pass (fused into the store)
    "thumbref_yuvj_inscale": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuvj420p", 1920, 1080, 3110400 + 86400),   # nothing pins yuv420p: the scale's own context converts the range (column walk, RNG)
    "half_yuvj_inscale": ("scale=1920:1080,format=yuvj420p", 3840, 2160, 12441600 + 3110400),
    "portrait_yuvj_inscale": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=yuvj420p", 1920, 1080, 4095360),
    "pad_only": ("pad=1920:1280:-1:-1:color=black", 1920, 1080, 3110400 + 1920 * 1280 * 3 // 2),
    "crop_only": ("crop=1080:1080", 1920, 1080, 1080 * 1080 * 3),
    "cvt_rgb24": ("format=rgb24", 1920, 1080, 3110400 + 6220800),                          # conversion alone (yuv2rgb_c_24_rgb)
    "cvt_rgb24_portrait": ("format=rgb24", 1080, 1920, 3110400 + 6220800),
    "cvt_rgba": ("format=rgba", 1920, 1080, 3110400 + 8294400),
    "cvt_rgbin": ("format=yuv420p", 1920, 1080, 6220800 + 3110400),                                  # rgb24 -> yuv420p, no scale
    "cvt_rgbin_yuvj": ("format=yuvj420p", 1920, 1080, 6220800 + 3110400),                             # rgb24 -> yuvj420p, no scale: the converting context writes full range
    "cvt_rgbain": ("format=yuv420p", 1920, 1080, 8294400 + 3110400),                                 # rgba -> yuv420p, no scale
    "cvt_bgrin": ("format=yuv420p", 1920, 1080, 6220800 + 3110400),                                  # bgr24 -> yuv420p, no scale: swscale's special converter
    "web_rgb24": ("scale=720:1280,format=rgb24", 1080, 1920, 3110400 + 2764800),      # 1.5x shrink written as RGB
    "web_rgba": ("scale=720:1280,format=rgba", 1080, 1920, 3110400 + 3686400),
    "4k_rgb24": ("scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920,format=rgb24", 3840, 2160, 3939840 + 6220800),   # 1.125x shrink, 5 taps
    "half_rgb24": ("scale=1920:1080,format=rgb24", 3840, 2160, 12441600 + 6220800),      # 2x shrink, 8 taps
    "thumbref_rgb24": ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=rgb24", 1920, 1080, 3110400 + 172800),
}


IN_FMT = {"quarter_rgbin": "rgb24", "cvt_rgbin_yuvj": "rgb24", "cvt_rgbain": "rgba", "thumbref_rgbin_inctx": "rgb24", "thumbref_rgbain": "rgba", "thumbref_portrait_rgbin": "rgb24", "portrait_rgbin_inctx": "rgb24", "thumbref_rgbin": "rgb24", "thumbref_rgbin_yuvj": "rgb24", "portrait_rgbin_yuvj": "rgb24", "portrait_rgbin": "rgb24", "stretch_rgbin": "rgb24", "cvt_rgbin": "rgb24", "half_rgbin": "rgb24", "web_rgbin": "rgb24", "thumb_rgbin": "rgb24", "cvt_bgrin": "bgr24"}


def plan_for(case):
    """The plan of a sweep case (packed-RGB input where the case says so)."""
    vf, iw, ih, _ = CASES[case]
    return tts.Plan.from_vf(vf, iw, ih, in_fmt=IN_FMT.get(case, "yuv420p"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--case", default="portrait")
    ap.add_argument("--segs", default="0,16,32,48,64,128")
    ap.add_argument("--ns", default="16,8")
    ap.add_argument("--extra", default="", help="comma list of ENV=V1|V2 knobs to cross")
    ap.add_argument("--dst-align", default="0", help="comma list: destination rows padded to a multiple of this many bytes (0 = packed rows); planar output")
    args = ap.parse_args()
    vf, iw, ih, algo = CASES[args.case]
    knobs = {"TTS_SEG_ROWS": args.segs.split(","), "TTS_STRIP_N": args.ns.split(","), "DST_ALIGN": args.dst_align.split(",")}
    for e in filter(None, args.extra.split(",")):
        k, v = e.split("=")
        knobs[k] = v.split("|")
    names = list(knobs)
    combos = list(itertools.product(*[knobs[k] for k in names]))
    plans = {}
    src = None
    for combo in combos:
        for k, v in zip(names, combo):
            os.environ[k] = v
        plans[combo] = plan_for(args.case)
    p0 = plans[combos[0]]
    src = torch.randint(0, 256, (args.frames, p0.in_bytes), dtype=torch.uint8, device="cuda")
    ow, oh = p0.out_size
    up = lambda v, a: (v + a - 1) // a * a
    dstr = {a: ((ow, (ow + 1) // 2) if a == 0 else (up(ow, a), up((ow + 1) // 2, a))) for a in map(int, knobs["DST_ALIGN"])}
    dst = torch.empty((args.frames, max([p0.out_bytes] + [p0.strided_bytes("out", *d) for d in dstr.values()])), dtype=torch.uint8, device="cuda")   # (flat: frames lie strided_bytes apart)
    sstr = (iw, (iw + 1) // 2)
    times = {c: [] for c in combos}
    for rnd in range(args.rounds + 1):
        for combo in combos:
            for k, v in zip(names, combo):
                os.environ[k] = v
            p = plans[combo]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            al = int(combo[names.index("DST_ALIGN")])
            for _ in range(args.iters):
                if len(dstr) > 1 or al:
                    p.run_batch_strided(src, sstr, dst, dstr[al], args.frames)
                else:
                    p.run_batch(src, dst if dst.shape[1] == p0.out_bytes else dst.view(-1)[: args.frames * p0.out_bytes].view(args.frames, p0.out_bytes))
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[combo].append(e0.elapsed_time(e1) / args.iters)
    print("case %s frames %d  (ms per batch: median / min ; algorithmic GB/s at median ; %% of 8 TB/s)" % (args.case, args.frames))
    for combo in sorted(combos, key=lambda c: sorted(times[c])[len(times[c]) // 2]):
        t = sorted(times[combo])
        med = t[len(t) // 2]
        gbs = algo * args.frames / (med * 1e-3) / 1e9
        print("  %-60s %8.4f / %8.4f   %8.1f GB/s  %5.1f%%   %.0f fps" % (
            " ".join("%s=%s" % (k.replace("TTS_", ""), v) for k, v in zip(names, combo)), med, t[0], gbs, gbs / 80.0,
            args.frames / (med * 1e-3)))


if __name__ == "__main__":
    main()
