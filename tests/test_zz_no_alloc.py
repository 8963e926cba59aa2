"""tts_run_host_static (include/ttscale.h): the allocation-free variant of tts_run_host, staged through the
library's `static __device__` scratch.  Checked through the C ABI against tts_run_host, bit-exact, on two
calls in a row -- the second with a smaller input, so that bytes the first call left in the scratch would
show -- then on a batch larger than the scratch (several passes) and on a frame that cannot fit."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

PORTRAIT = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
THUMB_REF = "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p"


def _both(tts, vf, iw, ih, n, pattern="random", seed=synth.SEED):
    import torch
    plan = tts.Plan.from_vf(vf, iw, ih)
    src = synth.frames(pattern, iw, ih, n, seed=seed)
    want = plan.run_host(src)
    got = np.full((n, plan.out_bytes), 0x5A, np.uint8)
    plan.run_host_static(src, got)
    torch.cuda.synchronize()
    return plan, got, want


def test_static_scratch_matches_run_host_twice_second_smaller(tts):
    _, got, want = _both(tts, PORTRAIT, 1920, 1080, 5)              # 31 MB through the 64 MiB scratch in one pass
    assert np.array_equal(got, want)
    _, got, want = _both(tts, THUMB_REF + ",format=yuvj420p", 720, 1280, 2, seed=7)   # smaller: pad + range kernels, other plan
    assert np.array_equal(got, want)
    _, got, want = _both(tts, "scale=96:54,format=rgb24", 64, 36, 1, pattern="zone")   # one tiny frame, packed RGB out
    assert np.array_equal(got, want)


def test_static_scratch_takes_large_batches_in_passes(tts, oracle):
    plan, got, want = _both(tts, PORTRAIT, 1920, 1080, 25)          # 155 MB: three passes over the scratch
    assert np.array_equal(got, want)
    assert np.array_equal(got[-1:], oracle.run_chain(plan.geom, synth.frames("random", 1920, 1080, 25)[-1:]))


def test_static_scratch_rejects_a_frame_that_cannot_fit(tts):
    plan = tts.Plan.from_vf("scale=7680:4320,format=rgba", 3840, 2160)   # 12.4 MB in + 132.7 MB out > 64 MiB
    src = np.zeros((1, plan.in_bytes), np.uint8)
    out = np.zeros((1, plan.out_bytes), np.uint8)
    with pytest.raises(tts.TtsError, match="UNSUPPORTED.*64 MiB"):
        plan.run_host_static(src, out)


def test_static_scratch_empty_and_bad_arguments(tts):
    plan = tts.Plan.from_vf("scale=96:54", 64, 36)
    assert tts._lib.lib().tts_run_host_static(plan._h, None, None, 0, None) == 0
    assert tts._lib.lib().tts_run_host_static(plan._h, None, None, 1, None) == -1
    assert tts._lib.lib().tts_run_host_static(None, None, None, 1, None) == -1
