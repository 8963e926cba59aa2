"""tts_run_host_static (include/ttscale.h): the allocation-free variant of tts_run_host, staged through the
library's `static __device__` scratch.  Checked through the C ABI against tts_run_host, bit-exact, on two
calls in a row -- the second with a smaller input, so that bytes the first call left in the scratch would
show -- then on a batch larger than the scratch (several passes) and on a frame that cannot fit.  The property in
the name is checked too: device memory in use is the same before and after a call (hipMemGetInfo), and the call
runs under stream capture -- where an allocation, a stream creation or a synchronisation would fail the capture
-- and replays from a graph."""
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


def test_static_scratch_allocates_nothing(tts):
    """Free device memory (hipMemGetInfo) is the same before and after a call: the staging is the library's static
    array, its address and the launch facts were looked up when the plan was made."""
    import torch
    plan = tts.Plan.from_vf(PORTRAIT, 1920, 1080)
    hsrc = torch.from_numpy(synth.frames("random", 1920, 1080, 4)).pin_memory()
    hout = torch.empty((4, plan.out_bytes), dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    plan.run_host_static(hsrc.numpy(), hout.numpy())                # the plan's FIRST call
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    plan.run_host_static(hsrc.numpy()[:2], hout.numpy()[:2])        # a second, smaller one
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    assert free1 == free0 and free2 == free0, (free0, free1, free2)
    assert np.array_equal(hout.numpy(), plan.run_host(hsrc.numpy()))
    # the Python wrapper refuses what it would have to copy (the call returns before its copies run) and partial frames
    with pytest.raises(tts.TtsError):
        plan.run_host_static(hsrc.numpy()[:, ::2], hout.numpy())
    with pytest.raises(tts.TtsError):
        plan.run_host_static(hsrc.numpy().ravel()[:-1], hout.numpy())


def test_static_scratch_under_stream_capture(tts, oracle):
    """The caller this entry point is for: captured into a HIP graph (pinned host buffers; copies, launches, nothing
    else) and replayed with new frames in the same buffers."""
    import torch
    iw, ih, n = 320, 180, 3
    plan = tts.Plan.from_vf("scale=180:320:force_original_aspect_ratio=increase,crop=180:320", iw, ih)
    a = synth.frames("random", iw, ih, n)
    hsrc = torch.empty((n, plan.in_bytes), dtype=torch.uint8).pin_memory()
    hout = torch.empty((n, plan.out_bytes), dtype=torch.uint8).pin_memory()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        plan.run_host_static(hsrc.numpy(), hout.numpy(), stream=torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        a2 = np.roll(a, rep, axis=0) ^ np.uint8(5 * rep)
        hsrc.numpy()[:] = a2
        hout.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(hout.numpy(), oracle.run_chain(plan.geom, a2)), rep
