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


@dataclass(frozen=True)
class Piece:
    """A run of consecutive frames of one clip: what a rank is handed when long clips are cut at GOP boundaries."""
    clip: int          # index into the clip list
    first: int
    frames: int


def split_gops(clips: Sequence[Clip], world: int, out_size: Tuple[int, int], gop: int = 30) -> List[Piece]:
    """Cut every clip that costs more than the mean rank load into near-equal pieces at GOP boundaries
    (multiples of `gop` frames: where a decoder can start), so that one long clip -- a 600-frame 4K clip
    among short ones, or fewer clips than GPUs -- spreads over several GPUs.  Frames are independent, so any
    frame boundary would do for the scaler; the GOP grid is what the decoder feeding it needs.  Shorter clips
    stay whole (a piece per clip).  Deterministic."""
    if world < 1 or gop < 1:
        raise ValueError("world and gop must be >= 1")
    total = sum(c.cost(*out_size) for c in clips)
    mean = total / world if world else 0
    pieces: List[Piece] = []
    for i, c in enumerate(clips):
        gops = (c.frames + gop - 1) // gop
        parts = 1
        if mean > 0 and c.cost(*out_size) > mean:
            # pieces of about a quarter of the mean load (the greedy assignment then evens the ranks out to a
            # few per cent), at most one piece per GOP
            parts = min(gops, -(-4 * c.cost(*out_size) // max(int(mean), 1)))
        first = 0
        for k in range(parts):
            g = gops // parts + (1 if k < gops % parts else 0)            # GOPs of this piece
            n = min(g * gop, c.frames - first)
            if n > 0:
                pieces.append(Piece(i, first, n))
            first += n
    return pieces


def assign_pieces(clips: Sequence[Clip], world: int, out_size: Tuple[int, int], gop: int = 30) -> List[List[Piece]]:
    """assign() over GOP-granular pieces: per rank, the pieces it processes (longest-processing-time first)."""
    pieces = split_gops(clips, world, out_size, gop)
    cost = [Clip(clips[p.clip].name, clips[p.clip].width, clips[p.clip].height, p.frames).cost(*out_size) for p in pieces]
    order = sorted(range(len(pieces)), key=lambda i: (-cost[i], i))
    load = [0] * world
    shards: List[List[Piece]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(pieces[i])
        load[r] += cost[i]
    return shards


def piece_imbalance(clips: Sequence[Clip], shards: List[List[Piece]], out_size: Tuple[int, int]) -> float:
    loads = [sum(Clip("", clips[p.clip].width, clips[p.clip].height, p.frames).cost(*out_size) for p in s) for s in shards]
    mean = sum(loads) / max(len(loads), 1)
    return max(loads) / mean if mean else 1.0


def imbalance(clips: Sequence[Clip], shards: List[List[int]], out_size: Tuple[int, int]) -> float:
    """max rank load / mean rank load (1.0 = perfect)."""
    loads = [sum(clips[i].cost(*out_size) for i in s) for s in shards]
    mean = sum(loads) / max(len(loads), 1)
    return max(loads) / mean if mean else 1.0


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/nodeN/cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(pci_bus_id: str, sysfs: str = "/sys") -> List[int]:
    """Keep this rank's host threads on the NUMA node its GPU hangs off: what matters for the
    host path (tts_run_host stages through pinned buffers that should be local to the GPU's root complex), not
    for HBM-resident batches.  `pci_bus_id` as torch.cuda.get_device_properties(i).pci_bus_id / hipDeviceGetPCIBusId
    print it ('0000:05:00.0').  Returns the CPUs the process was bound to, or [] when the machine does not say
    (no sysfs entry, node -1, not permitted): never an error -- an unpinned rank is only slower."""
    import os
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return []
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as f:
            cpus = parse_cpulist(f.read())
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return []
        os.sched_setaffinity(0, allowed)
        return allowed
    except (OSError, ValueError):
        return []


def gpu_pci_address(index: int) -> str:
    """'dddd:bb:dd.0' of visible GPU `index` from torch's device properties ('' if this torch does not tell)."""
    import torch
    pr = torch.cuda.get_device_properties(index)
    bus, dev, dom = getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None), getattr(pr, "pci_domain_id", 0)
    if isinstance(bus, str):
        return bus
    if bus is None or dev is None:
        return ""
    return "%04x:%02x:%02x.0" % (dom or 0, bus, dev)


def _checksum(out):
    """Sum of a batch's output bytes, as a 0-dim int64 CUDA tensor.  torch reduces uint8 to int64 at a few hundred GB/s
    (2.3 ms for 150 portrait frames, fifty times the scaler's own time); here the bytes are added four at a time inside
    their dwords first (two 16-bit lanes, then one: at most 1020, nothing carries), so the reduction runs over a
    quarter of the elements, as int32."""
    import torch
    flat = out.reshape(-1)
    n4 = flat.numel() // 4 * 4
    w = flat[:n4].view(torch.int32)
    t = (w & 0x00FF00FF) + ((w >> 8) & 0x00FF00FF)            # lanes: b0 + b1, b2 + b3
    total = ((t & 0xFFFF) + (t >> 16)).sum(dtype=torch.int64)
    if n4 != flat.numel():
        total = total + flat[n4:].sum(dtype=torch.int64)
    return total


def run_shard(clips: Sequence[Clip], my: Sequence, vf: str, make_frames, batch: int = 64, timing: Dict = None, checksum: bool = True) -> Dict[str, int]:
    """Process this rank's share on the current GPU: `my` holds clip indices (assign) or Pieces
    (assign_pieces).  Returns {clip name: checksum of the output bytes this rank produced for it} -- the sum
    of all output bytes, so the checksums of a clip's pieces add up to the whole clip's whatever the cut.
    `make_frames(clip, first, n)` supplies packed I420 frames as a uint8 CUDA tensor [n, in_bytes].
    `timing`, if given, receives "scale_s" (device time of the scaler launches alone, HIP events) and "frames"."""
    import torch
    from .scale import Plan
    plans: Dict[Tuple[int, int], Plan] = {}
    sums: Dict[str, int] = {}
    frames = 0
    events, totals = [], []
    for item in my:
        piece = item if isinstance(item, Piece) else Piece(item, 0, clips[item].frames)
        c = clips[piece.clip]
        plan = plans.get((c.width, c.height))
        if plan is None:                                   # plan switch: one plan per source geometry per GPU
            plan = plans[(c.width, c.height)] = Plan.from_vf(vf, c.width, c.height)
        # (nothing in the loop waits for the GPU: the checksum is accumulated on the device and the events are read when
        # the whole share has been queued, so a rank's wall time is its launches, not a round trip per batch)
        total = torch.zeros((), dtype=torch.int64, device="cuda")
        for first in range(piece.first, piece.first + piece.frames, batch):
            n = min(batch, piece.first + piece.frames - first)
            src = make_frames(c, first, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = plan.run_batch(src)
            e1.record()
            if checksum:                                   # (False: launches only -- torch's reductions take ten times the scaler's time)
                total += _checksum(out)
            events.append((e0, e1))
            frames += n
        totals.append((c.name, total))
    torch.cuda.synchronize()
    for name, total in totals:
        sums[name] = sums.get(name, 0) + int(total.item())
    scale_ms = sum(e0.elapsed_time(e1) for e0, e1 in events)
    if timing is not None:
        timing["scale_s"] = timing.get("scale_s", 0.0) + scale_ms * 1e-3
        timing["frames"] = timing.get("frames", 0) + frames
    return sums
