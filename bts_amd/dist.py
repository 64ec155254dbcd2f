"""Batch-sharded multi-GPU inference: one process per GPU, torch.distributed ('nccl' == RCCL on ROCm).

The reference shards with nn.DataParallel (bts_test.py:91), which re-broadcasts all parameters and
gathers all six outputs to GPU 0 on EVERY forward.  Frames are independent in eval mode
(bts.py:223-293 has no cross-sample op), so here: parameters are broadcast ONCE at start-up in a few
large flat buckets (xGMI is point-to-point; few large messages beat 600 small ones), every rank runs
its contiguous block of the batch, and only the five 1-channel depth maps are all-gathered (iconv1,
32 channels at full resolution, stays sharded: bts_test.py:133 discards it).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

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


def all_gather_depths(outs: Sequence[torch.Tensor], n_maps: int = 5, async_op: bool = False):
    """Gather the first ``n_maps`` outputs ([b,1,H,W] each) of every rank: returns ([world,n_maps,b,1,H,W], work).
    One packed buffer -> ONE collective per step instead of five."""
    packed = torch.stack([o for o in outs[:n_maps]], dim=0).contiguous()
    if not dist.is_initialized():
        return packed.unsqueeze(0), None
    world = dist.get_world_size()
    gathered = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    work = dist.all_gather_into_tensor(gathered.view(-1), packed.view(-1), async_op=async_op)
    return gathered, work


def unshard_depths(gathered: torch.Tensor) -> List[torch.Tensor]:
    """[world,n_maps,b,1,H,W] -> n_maps tensors [world*b,1,H,W] in global batch order."""
    world, n_maps, b = gathered.shape[:3]
    return [gathered[:, i].reshape((world * b,) + tuple(gathered.shape[3:])) for i in range(n_maps)]


def all_reduce_abs_min(decoder, async_op: bool = False):
    """Global ``abs_min`` of the three LPG layers over all ranks (a 3-float all-reduce(min), only when
    a caller wants the batch-wide diagnostic the reference logs, bts_main.py:484-486): updates
    ``decoder.lpg{8x8,4x4,2x2}.abs_min`` in place and returns (the [3] tensor, work)."""
    am = torch.stack([decoder.lpg8x8.abs_min, decoder.lpg4x4.abs_min, decoder.lpg2x2.abs_min]).float()
    work = None
    if dist.is_initialized():
        work = dist.all_reduce(am, op=dist.ReduceOp.MIN, async_op=async_op)
        if work is not None and not async_op:
            work = None
    decoder.lpg8x8.abs_min, decoder.lpg4x4.abs_min, decoder.lpg2x2.abs_min = am[0], am[1], am[2]
    return am, work


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
