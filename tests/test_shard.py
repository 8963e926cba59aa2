"""Multi-GPU path = clip sharding with no collective.  CPU tests: the assignment is
deterministic, disjoint, complete and balanced; a world_size-2 gloo run shows every rank derives
the same table on its own and that gathering per-clip results works."""
import os
import socket
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiktoken_amd.shard import Clip, Piece, assign, assign_pieces, imbalance, piece_imbalance, split_gops  # noqa: E402

OUT = (1080, 1920)


def corpus():
    # BASELINE.json configs[4]: 100 clips, 34 x 720p / 33 x 1080p / 33 x 4K, 300 frames each
    clips = [Clip("c%03d" % i, *((1280, 720) if i % 3 == 0 else (1920, 1080) if i % 3 == 1 else (3840, 2160)), 300)
             for i in range(100)]
    return clips


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_assignment_is_a_balanced_partition(world):
    clips = corpus()
    shards = assign(clips, world, OUT)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(len(clips)))
    assert imbalance(clips, shards, OUT) < 1.05
    assert assign(clips, world, OUT) == shards          # deterministic


def test_ragged_and_empty_inputs():
    assert assign([], 4, OUT) == [[], [], [], []]
    one = [Clip("only", 1920, 1080, 7)]
    assert assign(one, 8, OUT)[0] == [0] and sum(len(s) for s in assign(one, 8, OUT)) == 1
    with pytest.raises(ValueError):
        assign(one, 0, OUT)


def test_gop_split_spreads_long_clips():
    # BASELINE.json configs[3] short of GPUs' worth of clips: 3 x 600-frame 4K clips on 8 GPUs; clip-granular
    # sharding leaves 5 GPUs idle, GOP-granular pieces fill all 8
    clips = [Clip("k%d" % i, 3840, 2160, 600) for i in range(3)]
    assert imbalance(clips, assign(clips, 8, OUT), OUT) > 2.5
    shards = assign_pieces(clips, 8, OUT, gop=30)
    assert all(shards) and piece_imbalance(clips, shards, OUT) < 1.15
    for c in range(3):                                  # every clip: its pieces tile [0, 600) on the GOP grid
        ps = sorted((p for s in shards for p in s if p.clip == c), key=lambda p: p.first)
        assert ps[0].first == 0 and sum(p.frames for p in ps) == 600
        assert all(a.first + a.frames == b.first for a, b in zip(ps, ps[1:]))
        assert all(p.first % 30 == 0 for p in ps)
    assert assign_pieces(clips, 8, OUT, gop=30) == shards            # deterministic
    # one long clip among short ones: only the long one is cut
    mixed = [Clip("long", 3840, 2160, 6000)] + [Clip("s%d" % i, 1280, 720, 300) for i in range(20)]
    pieces = split_gops(mixed, 4, OUT, gop=60)
    assert sum(1 for p in pieces if p.clip == 0) >= 4 and all(p.frames == 300 for p in pieces if p.clip)
    assert piece_imbalance(mixed, assign_pieces(mixed, 4, OUT, gop=60), OUT) < 1.1
    # ragged: a clip shorter than a GOP, a clip that is not a whole number of GOPs, an empty list
    assert split_gops([Clip("t", 1920, 1080, 7)], 8, OUT, gop=30) == [Piece(0, 0, 7)]
    ps = split_gops([Clip("r", 3840, 2160, 100)], 4, OUT, gop=30)
    assert [p.first for p in ps] == [0, 30, 60, 90] and ps[-1].frames == 10
    assert split_gops([], 4, OUT) == [] and assign_pieces([], 2, OUT) == [[], []]
    with pytest.raises(ValueError):
        split_gops(clips, 0, OUT)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_mixed_corpus_pieces_partition(world):
    clips = corpus()
    shards = assign_pieces(clips, world, OUT)
    seen = sorted((p.clip, p.first, p.frames) for s in shards for p in s)
    assert [c for c, _, _ in seen] == list(range(100)) and all(f == 0 and n == 300 for _, f, n in seen)   # nothing here exceeds the mean: whole clips
    assert piece_imbalance(clips, shards, OUT) < 1.05


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clips = corpus()
    shards = assign(clips, world, OUT)
    mine = {clips[i].name: clips[i].frames * (rank + 1) for i in shards[rank]}     # stand-in per-clip result
    gathered = [None] * world
    dist.all_gather_object(gathered, (shards, mine))
    dist.destroy_process_group()
    q.put((rank, gathered))


def test_two_ranks_agree_without_talking():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tables = [g[0] for g in results[0]]
    assert tables[0] == tables[1], "ranks derived different shard tables"
    names = [n for g in results[0] for n in g[1]]
    assert sorted(names) == sorted(c.name for c in corpus())


@pytest.mark.gpu
def test_reduced_corpus_checksums_match_single_clip_runs():
    """A reduced BASELINE configs[4] corpus (2 + 2 + 2 clips at the real 720p / 1080p / 4K sizes) through
    run_shard on one GPU -- plan switches between clips included -- gives, per clip, the checksum of that clip
    run alone; cutting the clips into GOP pieces and running the pieces in shard order changes nothing."""
    import torch
    import tiktoken_amd as tts
    from tiktoken_amd.shard import run_shard
    vf = "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920"
    clips = [Clip("a720", 1280, 720, 9), Clip("b1080", 1920, 1080, 7), Clip("c4k", 3840, 2160, 5),
             Clip("d720", 1280, 720, 4), Clip("e1080", 1920, 1080, 11), Clip("f4k", 3840, 2160, 6)]

    def make_frames(c, first, n):
        nbytes = tts.frame_bytes(c.width, c.height)
        out = torch.empty((n, nbytes), dtype=torch.uint8, device="cuda")
        for k in range(n):                                        # a frame depends on (clip, frame number) only
            g = torch.Generator(device="cuda").manual_seed(zlib.crc32(("%s/%d" % (c.name, first + k)).encode()))
            out[k] = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda", generator=g)
        return out

    timing = {}
    whole = run_shard(clips, assign(clips, 1, OUT)[0], vf, make_frames, batch=4, timing=timing)
    assert timing["frames"] == sum(c.frames for c in clips) and timing["scale_s"] > 0
    for i, c in enumerate(clips):
        plan = tts.Plan.from_vf(vf, c.width, c.height)
        alone = int(plan.run_batch(make_frames(c, 0, c.frames)).sum(dtype=torch.int64).item())
        assert whole[c.name] == alone, c.name
    merged = {}
    for shard in assign_pieces(clips, 4, OUT, gop=2):             # 4 "ranks" worth of pieces, run one after another here
        for name, v in run_shard(clips, shard, vf, make_frames, batch=3).items():
            merged[name] = merged.get(name, 0) + v
    assert merged == whole


def test_numa_pinning_reads_sysfs_and_never_fails(tmp_path):
    """pin_to_gpu_numa: the GPU's NUMA node from a sysfs-shaped tree; anything missing means "not pinned"."""
    import os
    from tiktoken_amd.shard import parse_cpulist, pin_to_gpu_numa
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    try:
        assert pin_to_gpu_numa("0000:05:00.0", sysfs=str(tmp_path)) == []            # no such device
        dev = tmp_path / "bus/pci/devices/0000:05:00.0"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text("-1\n")
        assert pin_to_gpu_numa("0000:05:00.0", sysfs=str(tmp_path)) == []            # the platform does not say
        (dev / "numa_node").write_text("1\n")
        node = tmp_path / "devices/system/node/node1"
        node.mkdir(parents=True)
        first = min(before)
        (node / "cpulist").write_text("%d,100000-100001\n" % first)
        assert pin_to_gpu_numa("0000:05:00.0", sysfs=str(tmp_path)) == [first]
        assert os.sched_getaffinity(0) == {first}
    finally:
        os.sched_setaffinity(0, before)
