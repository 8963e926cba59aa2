"""tiktoken_amd: MI355X-native yuv420p scale / crop / pad for the forestrebel/tiktoken video path.

Only the hot path lives here: csrc/ (HIP kernels + C ABI, built into
libttscale.so) and this thin host-side mirror of the reference's "-vf string" interface."""
from ._lib import Geom, Frame, TtsError, LIB_PATH, SYMBOLS  # noqa: F401
from .scale import Plan, parse_vf, frame_bytes, build_filter, trim_range, loop_order, ALG  # noqa: F401

__all__ = ["Plan", "parse_vf", "frame_bytes", "build_filter", "trim_range", "loop_order", "Geom", "Frame", "TtsError", "ALG", "LIB_PATH", "SYMBOLS"]
