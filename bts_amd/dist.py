"""Batch-sharded multi-GPU inference: one process per GPU, torch.distributed ('nccl' == RCCL on ROCm).

The reference shards with nn.DataParallel (bts_test.py:91), which re-broadcasts all parameters and
gathers all six outputs to GPU 0 on EVERY forward.  Frames are independent in eval mode
(bts.py:223-293 has no cross-sample op), so here: parameters are broadcast ONCE at start-up in a few
large flat buckets (xGMI is point-to-point; few large messages beat 600 small ones), every rank runs
its contiguous block of the batch, and only the five 1-channel depth maps are all-gathered (iconv1,
32 channels at full resolution, stays sharded: bts_test.py:133 discards it).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition (DataParallel.scatter semantics): first total%world ranks get one more."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_module(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 64 << 20):
    """Broadcast parameters + buffers from ``src`` in flat buckets (one-time start-up cost).

    The received values are copied into the parameters THEMSELVES under ``no_grad`` (not through ``.data``), so every
    tensor's ``_version`` bumps and the packed-weight caches / captured graphs keyed on (data_ptr, _version) rebuild:
    a rank that ran a warm-up forward before the broadcast does not keep computing with its pre-broadcast packs."""
    if not dist.is_initialized():
        return
    tensors = list(module.parameters()) + list(module.buffers())
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype, ts in by_dtype.items():
            bucket, size = [], 0
            for t in ts + [None]:
                if t is not None:
                    bucket.append(t)
                    size += t.numel() * t.element_size()
                if bucket and (t is None or size >= bucket_bytes):
                    flat = torch.cat([x.detach().reshape(-1) for x in bucket])
                    dist.broadcast(flat, src=src)
                    off = 0
                    for x in bucket:
                        x.copy_(flat[off:off + x.numel()].view_as(x))
                        off += x.numel()
                    bucket, size = [], 0


class DepthGather:
    """Persistent buffers for the per-step all-gather of the ``n_maps`` one-channel depth maps ([b,1,H,W] each).

    ``packed[slot]`` is [n_maps, b_max, 1, H, W]; ``gathered[slot]`` is [world, n_maps, b_max, 1, H, W].  Nothing is
    allocated per step.  ``outputs(slot, b)`` hands out the maps of one step as contiguous views of ``packed[slot]``:
    a model told to write its results there (``BtsModel.output_buffers``) needs no pack copy at all -- LPG / get_depth
    store straight into the send buffer.  ``slots`` > 1 lets step i+1 compute into another slot while the collective of
    step i is still reading its own (the caller waits for the work of step i before it reuses slot i % slots).

    Uneven shards (a global batch that the ranks do not divide: the reference's evaluation sampler hands out
    ``indices[rank::world]``, distributed_sampler_no_evenly_divisible.py:62, and DataParallel.scatter chunks
    unevenly): every rank sends ``b_max`` = ceil(G / world) frames -- the fixed-size all_gather_into_tensor needs equal
    messages -- and ``maps(slot, global_batch)`` trims the pad frames of the short ranks (``shard_range`` order)."""

    def __init__(self, n_maps: int, b_max: int, H: int, W: int, device, dtype=torch.float32, slots: int = 2):
        self.n_maps, self.b_max, self.H, self.W = int(n_maps), int(b_max), int(H), int(W)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.slots = max(1, int(slots))
        shape = (self.n_maps, self.b_max, 1, self.H, self.W)
        self.packed = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(self.slots)]
        self.gathered = [torch.empty((self.world,) + shape, dtype=dtype, device=device) for _ in range(self.slots)]

    def outputs(self, slot: int, b: Optional[int] = None) -> List[torch.Tensor]:
        """The ``n_maps`` [b,1,H,W] result tensors of one step, as views of the send buffer of ``slot``."""
        b = self.b_max if b is None else int(b)
        if not 0 < b <= self.b_max:
            raise ValueError("DepthGather.outputs: %d frames do not fit the %d-frame buffer" % (b, self.b_max))
        return [self.packed[slot % self.slots][i, :b] for i in range(self.n_maps)]

    def pack(self, slot: int, outs: Sequence[torch.Tensor]):
        """Copy ``outs`` (the first ``n_maps`` of them) into the send buffer unless they already live there."""
        dst = self.outputs(slot, outs[0].shape[0])
        for d, o in zip(dst, outs[:self.n_maps]):
            if o.data_ptr() != d.data_ptr():
                d.copy_(o)

    def gather(self, slot: int, async_op: bool = False):
        """ONE collective for all maps of the step (None when torch.distributed is not initialised)."""
        slot %= self.slots
        if not dist.is_initialized():
            self.gathered[slot][0].copy_(self.packed[slot])
            return None
        return dist.all_gather_into_tensor(self.gathered[slot].view(-1), self.packed[slot].view(-1), async_op=async_op)

    def maps(self, slot: int, global_batch: Optional[int] = None) -> List[torch.Tensor]:
        """``n_maps`` tensors [G,1,H,W] in global batch order (views when the shards are even)."""
        return unshard_depths(self.gathered[slot % self.slots], global_batch)


_gather_cache = {}


def all_gather_depths(outs: Sequence[torch.Tensor], n_maps: int = 5, async_op: bool = False,
                      global_batch: Optional[int] = None):
    """Gather the first ``n_maps`` outputs ([b,1,H,W] each) of every rank: returns ([world,n_maps,b_max,1,H,W], work).
    One packed buffer -> ONE collective per step instead of five.  Buffers are persistent (a ``DepthGather`` per
    (shape, device), two alternating slots: the result of a call stays valid until the second next call with that shape).
    ``global_batch``: total frames over all ranks when the shards may be uneven (``shard_range`` sizes): this rank's
    ``outs`` are zero-padded to ceil(global_batch / world) frames; pass the same value to ``unshard_depths``."""
    b, _, H, W = outs[0].shape
    world = dist.get_world_size() if dist.is_initialized() else 1
    b_max = b if global_batch is None else -(-int(global_batch) // world)
    if b > b_max:
        raise ValueError("all_gather_depths: %d local frames but global_batch %s over %d ranks allows %d" % (b, global_batch, world, b_max))
    key = (n_maps, b_max, H, W, str(outs[0].device), outs[0].dtype, world)
    ent = _gather_cache.get(key)
    if ent is None:
        if len(_gather_cache) >= 4:
            _gather_cache.pop(next(iter(_gather_cache)))
        ent = _gather_cache[key] = [DepthGather(n_maps, b_max, H, W, outs[0].device, outs[0].dtype, slots=2), 0]
    g, slot = ent[0], ent[1]
    ent[1] = (slot + 1) % g.slots
    g.pack(slot, outs)
    work = g.gather(slot, async_op=async_op)
    return g.gathered[slot], work


def unshard_depths(gathered: torch.Tensor, global_batch: Optional[int] = None) -> List[torch.Tensor]:
    """[world,n_maps,b_max,1,H,W] -> n_maps tensors [G,1,H,W] in global batch order.  ``global_batch`` (G) given: rank
    r contributed ``shard_range(G, r, world)`` frames, its pad frames are dropped."""
    world, n_maps, b = gathered.shape[:3]
    if global_batch is None or int(global_batch) == world * b:
        return [gathered[:, i].reshape((world * b,) + tuple(gathered.shape[3:])) for i in range(n_maps)]
    sizes = [hi - lo for lo, hi in (shard_range(int(global_batch), r, world) for r in range(world))]
    if max(sizes) > b:
        raise ValueError("unshard_depths: global_batch %d needs %d frames per rank, the buffer holds %d" % (global_batch, max(sizes), b))
    return [torch.cat([gathered[r, i, :sizes[r]] for r in range(world)], dim=0) for i in range(n_maps)]


def all_reduce_abs_min(decoder, async_op: bool = False):
    """Global ``abs_min`` of the three LPG layers over all ranks (a 3-float all-reduce(min), only when
    a caller wants the batch-wide diagnostic the reference logs, bts_main.py:484-486): updates
    ``decoder.lpg{8x8,4x4,2x2}.abs_min`` in place and returns (the [3] tensor, work).

    NaN: the reference's ``torch.abs(divided).min()`` (bts.py:167) is NaN as soon as one denominator is, and the log line
    exists to hunt NaNs -- but ``ReduceOp.MIN`` over ranks leaves NaN handling to the backend.  ``abs_min`` is never
    negative, so a rank's NaN travels as -1 (which wins every MIN) and comes back as NaN: one NaN rank makes the global
    value NaN on every rank, whatever the backend does with NaN operands.  With ``async_op`` the decode happens in
    ``work.wait()``'s caller: use ``decode_abs_min`` on the returned tensor after waiting."""
    am = torch.stack([decoder.lpg8x8.abs_min, decoder.lpg4x4.abs_min, decoder.lpg2x2.abs_min]).float()
    am = torch.where(torch.isnan(am), torch.full_like(am, -1.0), am)
    work = None
    if dist.is_initialized():
        work = dist.all_reduce(am, op=dist.ReduceOp.MIN, async_op=async_op)
        if work is not None and not async_op:
            work = None
    if work is None:
        am = decode_abs_min(am)
        decoder.lpg8x8.abs_min, decoder.lpg4x4.abs_min, decoder.lpg2x2.abs_min = am[0], am[1], am[2]
    return am, work


def decode_abs_min(am: torch.Tensor) -> torch.Tensor:
    """Undo ``all_reduce_abs_min``'s NaN encoding (-1 -> NaN)."""
    return torch.where(am < 0, torch.full_like(am, float("nan")), am)


def shard_indices(total: int, rank: int, world: int) -> List[int]:
    """Dataset sharding for evaluation without padding: rank r takes indices r, r+world, ... (what the reference's
    DistributedSamplerNoEvenlyDivisible yields, distributed_sampler_no_evenly_divisible.py:62) -- ranks may differ by
    one sample, nothing is duplicated."""
    return list(range(rank, total, world))


def all_reduce_eval_measures(measures: torch.Tensor) -> torch.Tensor:
    """Sum the per-rank evaluation accumulator over all ranks (the reference's online eval keeps 9 error sums + a
    sample count and all-reduces them, bts_main.py:258-260); returns the tensor (reduced in place)."""
    if dist.is_initialized():
        dist.all_reduce(measures, op=dist.ReduceOp.SUM)
    return measures
