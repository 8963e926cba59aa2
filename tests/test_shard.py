"""Multi-GPU path = clip sharding with no collective.  CPU tests: the assignment is
deterministic, disjoint, complete and balanced; a world_size-2 gloo run shows every rank derives
the same table on its own and that gathering per-clip results works."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiktoken_amd.shard import Clip, assign, imbalance  # noqa: E402

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
