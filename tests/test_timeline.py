"""Time-domain steps: trim and loop as frame orders.  No pixels are computed by these
steps in the reference (scripts/loopVideos.js:54-100 extracts three segments and concatenates them; its
`crossfade` setting is never read), so parity here is an index list, checked against the oracle's
per-frame timestamp test, plus the scaler consuming that list on the GPU."""
import numpy as np
import pytest

import synth


@pytest.mark.parametrize("n,fps,start,dur,repeats", [
    (312, 30, 3, 2, 4),        # birds_demo.mp4: 10.4 s, loopStrategy of scripts/loopVideos.js:12-22
    (363, 30, 8, 2, 3),        # forest_demo.mp4: 12.1 s, :23-33
    (100, 25, 1.5, 0.5, 2),
    (90, 30000 / 1001, 1, 1, 3),   # NTSC rate: segment edges fall between frames
    (48, 24, 0, 1, 2),         # no initial segment
    (48, 24, 1, 5, 2),         # loop segment runs to the end: no end segment
    (48, 24, 3, 1, 1),         # one repeat: the clip itself
    (48, 24, 1, 1, 0),         # segment dropped
    (10, 30, 5, 1, 3),         # start beyond the end: nothing to repeat
    (0, 30, 0, 1, 2),
])
def test_loop_order_matches_oracle(tts, oracle, n, fps, start, dur, repeats):
    got = tts.loop_order(n, fps, start, dur, repeats)
    assert got.tolist() == oracle.loop_order(n, fps, start, dur, repeats)


def test_loop_lengths_the_reference_asks_for(tts):
    """scripts/loopVideos.js:14,19,25,30: the loops must take the clips past 15 s."""
    birds = tts.loop_order(312, 30, 3, 2, 4)
    forest = tts.loop_order(363, 30, 8, 2, 3)
    assert len(birds) == 312 + 3 * 60 and len(birds) / 30 > 15
    assert len(forest) == 363 + 2 * 60 and len(forest) / 30 > 15
    assert birds[:90].tolist() == list(range(90)) and birds[90:150].tolist() == birds[150:210].tolist() == list(range(90, 150))
    assert birds[-1] == 311


@pytest.mark.parametrize("n,fps,start,dur", [(300, 30, 0, 5), (300, 30, 2.5, None), (90, 30000 / 1001, 0.5, 1.25), (10, 30, 1, 5), (10, 30, 0, 0)])
def test_trim_range_matches_oracle(tts, oracle, n, fps, start, dur):
    first, count = tts.trim_range(n, fps, start, dur)
    want = oracle.segment_frames(n, fps, start, dur)
    assert list(range(first, first + count)) == want


def test_bad_arguments(tts):
    with pytest.raises(tts.TtsError):
        tts.loop_order(10, 30, -1, 1, 2)
    with pytest.raises(tts.TtsError):
        tts.loop_order(10, 30, 1, 1, -2)
    with pytest.raises(tts.TtsError):
        tts.loop_order(10, 0, 1, 1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("vf,iw,ih", [("scale=w=720:h=1280", 360, 640), ("format=yuv420p", 64, 36),
                                      ("scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p", 720, 1280)])
def test_scaler_consumes_a_loop_order(tts, oracle, vf, iw, ih):
    """A looped clip scaled straight from the original frames == the oracle on the frames gathered first."""
    import torch
    n, fps = 24, 6
    src = synth.frames("random", iw, ih, n)
    order = tts.loop_order(n, fps, 1, 1, 3)
    assert len(order) == n + 2 * 6
    plan = tts.Plan.from_vf(vf, iw, ih)
    got = plan.run_batch_order(torch.from_numpy(src).cuda(), order).cpu().numpy()
    assert np.array_equal(got, oracle.run_chain(plan.geom, src[order]))
    # any order at all, repeats and reversals included
    rnd = np.random.default_rng(3).integers(0, n, 40).astype(np.int32)
    got = plan.run_batch_order(torch.from_numpy(src).cuda(), rnd).cpu().numpy()
    assert np.array_equal(got, oracle.run_chain(plan.geom, src[rnd]))
    with pytest.raises(tts.TtsError):
        plan.run_batch_order(torch.from_numpy(src).cuda(), np.array([0, n], np.int32))


def test_trim_range_takes_infinite_and_absurd_times(tts):
    """-ss / -t far past the clip (or infinite): an empty range at the clip's end, never an out-of-range cast."""
    for start, dur in ((float("inf"), 1.0), (1e300, 1e300), (0.0, float("inf")), (3.0, 1e308)):
        first, count = tts.trim_range(100, 30, start, dur)
        assert 0 <= first <= 100 and 0 <= count <= 100 - first
    assert tts.trim_range(100, 30, 0.0, float("inf")) == (0, 100)
