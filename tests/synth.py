"""Deterministic synthetic yuv420p frames, seed 0x5EED.  Packed I420, uint8."""
import numpy as np

SEED = 0x5EED
PATTERNS = ("random", "limited", "hramp", "vramp", "checker", "zone", "impulse", "zeros", "ones")


def frame_bytes(w, h):
    return w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)


def _plane(pattern, w, h, rng, chroma):
    yy, xx = np.mgrid[0:h, 0:w]
    if pattern == "random":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if pattern == "limited":
        return rng.integers(16, 241 if chroma else 236, (h, w), dtype=np.uint8)
    if pattern == "hramp":
        return ((xx * 255) // max(w - 1, 1)).astype(np.uint8)
    if pattern == "vramp":
        return ((yy * 255) // max(h - 1, 1)).astype(np.uint8)
    if pattern == "checker":
        return ((((xx >> 3) + (yy >> 3)) & 1) * 255).astype(np.uint8)
    if pattern == "zone":
        r2 = (xx - w / 2.0) ** 2 + (yy - h / 2.0) ** 2
        return (127.5 + 127.5 * np.cos(r2 * (np.pi / max(w, h)))).astype(np.uint8)
    if pattern == "impulse":   # single-pixel 0/255 columns and rows: forces overshoot clipping
        p = np.where((xx % 7 == 0) | (yy % 11 == 0), 255, 0).astype(np.uint8)
        p[:, ::13] = 0
        return p
    if pattern == "zeros":
        return np.zeros((h, w), np.uint8)
    if pattern == "ones":
        return np.full((h, w), 255, np.uint8)
    raise ValueError(pattern)


def frames(pattern, w, h, n=1, seed=SEED):
    """[n, frame_bytes(w, h)] uint8."""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    out = np.empty((n, frame_bytes(w, h)), np.uint8)
    for i in range(n):
        rng = np.random.default_rng([seed, i])
        y = _plane(pattern, w, h, rng, False)
        u = _plane(pattern, cw, ch, rng, True)
        v = _plane(pattern, cw, ch, rng, True)[::-1, ::-1] if pattern not in ("random", "limited") else _plane(pattern, cw, ch, rng, True)
        out[i] = np.concatenate([y.ravel(), u.ravel(), np.ascontiguousarray(v).ravel()])
    return out


def split(frame, w, h):
    cw, ch = (w + 1) // 2, (h + 1) // 2
    y = frame[: w * h].reshape(h, w)
    u = frame[w * h: w * h + cw * ch].reshape(ch, cw)
    v = frame[w * h + cw * ch:].reshape(ch, cw)
    return y, u, v
