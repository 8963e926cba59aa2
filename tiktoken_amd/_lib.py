"""ctypes binding of include/ttscale.h.  There is no CPU fallback: if the HIP library is
missing this raises, loudly, and nothing else in the package works."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libttscale.so")
if os.environ.get("TTS_LIB"):          # tools only: an experiments build (make -C tiktoken_amd/csrc exp)
    LIB_PATH = os.environ["TTS_LIB"]


class Geom(ctypes.Structure):
    """tts_geom (include/ttscale.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "in_w", "in_h", "pre_x", "pre_y", "pre_w", "pre_h", "do_scale", "alg", "sc_w", "sc_h",
        "crop_x", "crop_y", "crop_w", "crop_h", "do_pad", "pad_x", "pad_y", "out_w", "out_h",
        "fill_y", "fill_u", "fill_v", "out_range", "out_fmt", "rgb_stage", "fill_rgb", "in_fmt", "sc_range")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Frame(ctypes.Structure):
    """tts_frame (include/ttscale.h)."""
    _fields_ = [("plane", ctypes.c_void_p * 3), ("stride", ctypes.c_int32 * 3),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("mem", ctypes.c_int32)]


# every symbol include/ttscale.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "tts_parse_vf": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Geom)]),
    "tts_parse_vf_fmt": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Geom)]),
    "tts_plan_create": (ctypes.c_int, [ctypes.POINTER(Geom), ctypes.POINTER(ctypes.c_void_p)]),
    "tts_plan_destroy": (None, [ctypes.c_void_p]),
    "tts_plan_geom": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Geom)]),
    "tts_plan_device": (ctypes.c_int, [ctypes.c_void_p]),
    "tts_plan_frame_bytes": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_int]),
    "tts_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Frame), ctypes.POINTER(Frame), ctypes.c_int, ctypes.c_void_p]),
    "tts_run_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tts_trim_range": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "tts_loop_order": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    "tts_run_batch_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "tts_run_batch_strided": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tts_run_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "tts_run_host_static": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "tts_trim": (None, []),
    "tts_plan_filter": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_void_p]),
    "tts_build_filter": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_void_p]),
    "tts_plan_kernel_kind": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "tts_plan_kernel_name": (ctypes.c_char_p, [ctypes.c_void_p, ctypes.c_int]),
    "tts_frame_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "tts_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "tts_last_error": (ctypes.c_char_p, []),
    "tts_version": (ctypes.c_char_p, []),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "tiktoken_amd: %s is missing. Build it with `python -m tiktoken_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64 (same soname as the
        # system one libttscale links).  Load torch's first so device pointers, streams and
        # the current device are shared between torch and the library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class TtsError(RuntimeError):
    """Carries the status names the web spec expects in thrown messages
    (reference: web/src/tests/video/compression.test.js:62-90)."""

    def __init__(self, status, detail=""):
        self.status = status
        name = lib().tts_strerror(status).decode()
        super().__init__("%s%s" % (name, (": " + detail) if detail else ""))


def check(status, detail=""):
    if status != 0:
        raise TtsError(status, detail)
