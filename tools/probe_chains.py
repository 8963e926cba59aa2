import os, sys
sys.path.insert(0, os.getcwd())
import tiktoken_amd as tts
cases = [
 ("format=rgb24", 64, 47, "yuv420p"), ("format=rgb24", 63, 48, "yuv420p"), ("format=rgb24", 63, 47, "yuv420p"),
 ("scale=128:95,format=yuv420p,format=rgb24", 64, 48, "yuv420p"), ("scale=127:96,format=yuv420p,format=rgb24", 64, 48, "yuv420p"),
 ("scale=128:96,format=yuv420p,format=rgb24", 64, 48, "yuv420p"),
 ("crop=32:21:4:2,format=rgb24", 64, 48, "yuv420p"), ("pad=80:61:-1:-1,format=rgb24", 64, 47, "yuv420p"),
 ("format=yuvj420p,format=rgb24", 64, 48, "yuv420p"), ("scale=96:54,format=yuvj420p,format=rgb24", 64, 36, "yuv420p"),
 ("format=yuvj420p,scale=128:96,format=rgb24", 64, 48, "yuv420p"),
 ("scale=96:54,format=rgb24", 64, 36, "rgb24"), ("format=rgb24", 64, 36, "rgb24"), ("scale=48:27", 63, 36, "rgb24"), ("format=yuv420p", 63, 36, "rgb24"),
 ("pad=80:60", 64, 36, "rgb24"), ("scale=96:54,pad=128:72:-1:-1", 64, 36, "rgb24"), ("format=yuvj420p", 64, 36, "rgb24"), ("scale=96:54,format=yuvj420p", 64, 36, "rgb24"),
 ("scale=96:54", 64, 36, "bgr24"), ("scale=96:54", 64, 36, "rgba"), ("scale=96:54", 64, 36, "bgra"),
 ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 720, 1280, "rgb24"),
]
for vf, iw, ih, fmt in cases:
    try:
        p = tts.Plan.from_vf(vf, iw, ih, in_fmt=fmt)
        print("OK   %-8s %dx%d %-60s -> %s" % (fmt, iw, ih, vf, p.kernel_name(0)))
    except Exception as e:
        print("FAIL %-8s %dx%d %-60s -> %s" % (fmt, iw, ih, vf, str(e)[:110]))
