"""Clip sharding across GPUs: clips are independent, so the only multi-GPU
machinery is a deterministic assignment of clips to ranks -- no data-path collective, xGMI idle.

The reference processes its clips one after another in a sequential loop
(scripts/processVideos.js:167-169); here every rank takes the clips `assign()` gives it and runs them
through its own plans on its own GPU.  torch.distributed is only used by callers that want to
gather per-clip results (checksums, timings) at the end."""
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple


@dataclass(frozen=True)
class Clip:
    name: str
    width: int
    height: int
    frames: int

    def cost(self, out_w: int, out_h: int) -> int:
        """Relative processing cost: output bytes dominate the traffic, source bytes the gathers."""
        return self.frames * (out_w * out_h * 3 // 2 + self.width * self.height * 3 // 8)


def assign(clips: Sequence[Clip], world: int, out_size: Tuple[int, int]) -> List[List[int]]:
    """Greedy longest-processing-time-first: returns, per rank, the indices of its clips.
    Deterministic (ties broken by index) so every rank computes the same table without talking."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(clips)), key=lambda i: (-clips[i].cost(*out_size), i))
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += clips[i].cost(*out_size)
    return shards


def imbalance(clips: Sequence[Clip], shards: List[List[int]], out_size: Tuple[int, int]) -> float:
    """max rank load / mean rank load (1.0 = perfect)."""
    loads = [sum(clips[i].cost(*out_size) for i in s) for s in shards]
    mean = sum(loads) / max(len(loads), 1)
    return max(loads) / mean if mean else 1.0


def run_shard(clips: Sequence[Clip], my: Sequence[int], vf: str, make_frames, batch: int = 64) -> Dict[str, int]:
    """Process this rank's clips on the current GPU; returns {clip name: checksum of its output bytes}.
    `make_frames(clip, first, n)` supplies packed I420 frames as a uint8 CUDA tensor [n, in_bytes]."""
    import torch
    from .scale import Plan
    plans: Dict[Tuple[int, int], Plan] = {}
    sums: Dict[str, int] = {}
    for i in my:
        c = clips[i]
        plan = plans.get((c.width, c.height))
        if plan is None:
            plan = plans[(c.width, c.height)] = Plan.from_vf(vf, c.width, c.height)
        total = 0
        for first in range(0, c.frames, batch):
            n = min(batch, c.frames - first)
            out = plan.run_batch(make_frames(c, first, n))
            total += int(out.to(torch.int64).sum().item())
        sums[c.name] = total
    return sums
