"""ctypes wrapper around oracle/_build/libtto_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
PARITY UNPINNED: see the header of swscale_oracle.c."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libtto_oracle.so")

BICUBIC, AREA, BILINEAR = 0, 1, 2


class _Filter(ctypes.Structure):
    _fields_ = [("coef", ctypes.POINTER(ctypes.c_int16)), ("pos", ctypes.POINTER(ctypes.c_int32)),
                ("size", ctypes.c_int), ("dst", ctypes.c_int), ("src", ctypes.c_int)]


class Chain(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "in_w", "in_h", "pre_x", "pre_y", "pre_w", "pre_h", "do_scale", "alg", "sc_w", "sc_h",
        "crop_x", "crop_y", "crop_w", "crop_h", "do_pad", "pad_x", "pad_y", "out_w", "out_h",
        "fill_y", "fill_u", "fill_v", "out_range", "out_fmt", "rgb_stage", "fill_rgb", "in_fmt", "sc_range")]


_lib = None


def build():
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        L = ctypes.CDLL(SO)
        L.tto_filter_init.argtypes = [ctypes.POINTER(_Filter), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.tto_filter_free.argtypes = [ctypes.POINTER(_Filter)]
        L.tto_scale_plane.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.tto_scale_dims.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int)] * 2
        L.tto_crop_offset.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
        L.tto_run_chain.argtypes = [ctypes.POINTER(Chain)] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + \
                                   [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
        L.tto_run_chain_rgb.argtypes = [ctypes.POINTER(Chain)] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + \
                                       [ctypes.c_void_p, ctypes.c_int]
        L.tto_run_chain_rgbin.argtypes = [ctypes.POINTER(Chain), ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
        L.tto_sws_yuv420_to_rgb.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p] + [ctypes.c_int] * 5
        _lib = L
    return _lib


def filter_table(src, dst, one, alg=BICUBIC):
    f = _Filter()
    assert lib().tto_filter_init(ctypes.byref(f), src, dst, one, alg) == 0
    coef = np.ctypeslib.as_array(f.coef, (dst, f.size)).copy()
    pos = np.ctypeslib.as_array(f.pos, (dst,)).copy()
    lib().tto_filter_free(ctypes.byref(f))
    return coef, pos


def scale_plane(src, dw, dh, alg=BICUBIC):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    assert lib().tto_scale_plane(src.ctypes.data, sw, sw, sh, dst.ctypes.data, dw, dw, dh, alg) == 0
    return dst


def scale_dims(iw, ih, w, h, foar=0, fdb=1):
    ow, oh = ctypes.c_int(), ctypes.c_int()
    rc = lib().tto_scale_dims(iw, ih, w, h, foar, fdb, ctypes.byref(ow), ctypes.byref(oh))
    return (ow.value, oh.value) if rc == 0 else None


def crop_offset(inn, out, value=None):
    return lib().tto_crop_offset(inn, out, 0.0 if value is None else float(value), 0 if value is None else 1)


def chain_from(geom):
    """geom: anything with the tts_geom field names (dict or ctypes struct)."""
    c = Chain()
    for n, _ in Chain._fields_:
        setattr(c, n, geom.get(n, 0) if isinstance(geom, dict) else getattr(geom, n))
    return c


def frame_bytes(w, h):
    return w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)


class Unsupported(Exception):
    """swscale takes a path here that the oracle does not restate (see tto_sws_yuv420_to_rgb)."""


def out_frame_bytes(c):
    """Bytes of one output frame of a chain: packed I420, or out_w * out_h * bpp for packed RGB."""
    if c.out_fmt:
        return c.out_w * c.out_h * (4 if c.out_fmt >= 3 else 3)
    return frame_bytes(c.out_w, c.out_h)


def in_frame_bytes(c):
    return c.in_w * c.in_h * (4 if c.in_fmt >= 3 else 3) if c.in_fmt else frame_bytes(c.in_w, c.in_h)


def run_chain(geom, frames):
    """frames: uint8 [n, in_frame_bytes] (packed I420, or packed rgb24 when geom.in_fmt says so)
    -> [n, out_frame_bytes] (packed I420 or packed RGB)."""
    c = chain_from(geom)
    frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, in_frame_bytes(c))
    if c.in_fmt:
        out = np.empty((frames.shape[0], frame_bytes(c.out_w, c.out_h)), np.uint8)
        ow, oh = c.out_w, c.out_h
        ocw, och = (ow + 1) // 2, (oh + 1) // 2
        for i in range(frames.shape[0]):
            d = out[i].ctypes.data
            rc = lib().tto_run_chain_rgbin(ctypes.byref(c), frames[i].ctypes.data, c.in_w * (4 if c.in_fmt >= 3 else 3), d, d + ow * oh, d + ow * oh + ocw * och, ow, ocw, ocw)
            assert rc == 0, rc
        return out
    out = np.empty((frames.shape[0], out_frame_bytes(c)), np.uint8)
    iw, ih, ow, oh = c.in_w, c.in_h, c.out_w, c.out_h
    icw, ich, ocw, och = (iw + 1) // 2, (ih + 1) // 2, (ow + 1) // 2, (oh + 1) // 2
    if c.out_fmt:
        bpp = 4 if c.out_fmt >= 3 else 3
        for i in range(frames.shape[0]):
            s, d = frames[i].ctypes.data, out[i].ctypes.data
            rc = lib().tto_run_chain_rgb(ctypes.byref(c), s, s + iw * ih, s + iw * ih + icw * ich, iw, icw, icw, d, ow * bpp)
            if rc == -2:
                raise Unsupported("swscale path not restated for this geometry")
            assert rc == 0, rc
        return out
    for i in range(frames.shape[0]):
        s, d = frames[i].ctypes.data, out[i].ctypes.data
        rc = lib().tto_run_chain(ctypes.byref(c), s, s + iw * ih, s + iw * ih + icw * ich, iw, icw, icw,
                                 d, d + ow * oh, d + ow * oh + ocw * och, ow, ocw, ocw)
        assert rc == 0
    return out


# ---- time-domain steps (no pixels): which source frames a trimmed / looped clip holds ---------------

def segment_frames(n_frames, fps, start, duration=None):
    """Frames a `ffmpeg -ss start -i clip -t duration` extraction keeps (scripts/loopVideos.js:42-52,
    scripts/processVideos.js:84-92): those whose timestamp i/fps lies in [start, start+duration).
    Exact rational arithmetic, one frame at a time."""
    from fractions import Fraction
    fps, start = Fraction(fps).limit_denominator(1001 * 64), Fraction(start).limit_denominator(10 ** 6)
    end = None if duration is None else start + Fraction(duration).limit_denominator(10 ** 6)
    return [i for i in range(n_frames) if Fraction(i) / fps >= start and (end is None or Fraction(i) / fps < end)]


def loop_order(n_frames, fps, extract_start, extract_duration, repeats):
    """createLoopedVideo (scripts/loopVideos.js:54-100) as a list of source frame numbers: the three
    extractions of :62, :66, :73 and the concat list of :77-81."""
    loop = segment_frames(n_frames, fps, extract_start, extract_duration)
    initial = segment_frames(n_frames, fps, 0, extract_start)
    from fractions import Fraction
    input_duration = Fraction(n_frames) / Fraction(fps).limit_denominator(1001 * 64)
    end_duration = input_duration - (Fraction(extract_start).limit_denominator(10 ** 6) + Fraction(extract_duration).limit_denominator(10 ** 6))
    end = segment_frames(n_frames, fps, extract_start + extract_duration, float(end_duration)) if end_duration > 0 else []
    return initial + loop * repeats + end
