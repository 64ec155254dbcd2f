"""CPU, world_size 2, gloo: the N>1 host path of bench.py / bts_amd.dist -- contiguous batch sharding, the
one-time flat-bucket weight broadcast, and the packed all-gather of the five depth maps.  The per-rank
"model" here is the CPU oracle decoder (a test stand-in; the product forward needs a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bts_amd import dist as bdist
from bts_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                      # ranks start with DIFFERENT weights
        from oracle import bts_oracle as O
        feat = synth.ENCODER_CHANNELS["densenet161_bts"]
        # a small module tree with conv + BN (params, buffers incl. an int64 one) to broadcast
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
        with torch.no_grad():
            m[1].running_mean.normal_()
            m[1].num_batches_tracked.fill_(rank + 5)
        versions = [x._version for x in list(m.parameters()) + list(m.buffers())]
        bdist.broadcast_module(m, src=0, bucket_bytes=256)      # tiny buckets -> several collectives
        # the copy goes into the tensors themselves, so every _version bumps: packed-weight caches and captured graphs
        # keyed on (data_ptr, _version) rebuild after a broadcast that follows a warm-up forward
        assert all(x._version > v for x, v in zip(list(m.parameters()) + list(m.buffers()), versions))
        flat = torch.cat([t.detach().double().reshape(-1) for t in list(m.parameters()) + list(m.buffers())])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "weights differ after broadcast"

        # shard a global batch of 3 frames (uneven: 2 + 1) and run the decoder per rank
        GB, H, W = 3, 32, 64
        lo, hi = bdist.shard_range(GB, rank, world)
        feats = synth.encoder_features(feat, GB, H, W, seed=9)
        focal = synth.focal_values(GB, "kitti", seed=9)
        state = O.state_from_numpy(synth.decoder_state(feat, 512, 0))
        def run_frames(a, b):      # frame by frame: CPU conv algorithms may differ with batch size, frames may not
            per = []
            with torch.no_grad():
                for i in range(a, b):
                    fi = [None] + [torch.from_numpy(f[i:i + 1]) for f in feats[1:]]
                    per.append(O.decoder_forward(state, fi, torch.from_numpy(focal[i:i + 1]), 80.0, "kitti"))
            return [torch.cat([p[j] for p in per]) for j in range(6)]

        outs = run_frames(lo, hi)
        # uneven shards: pad to the max shard for the fixed-size all_gather_into_tensor, then trim
        bmax = (GB + world - 1) // world
        padded = [torch.cat([o, torch.zeros((bmax - o.shape[0],) + tuple(o.shape[1:]))]) for o in outs[:5]]
        gathered, work = bdist.all_gather_depths(padded, 5, async_op=True)
        work.wait()
        maps = bdist.unshard_depths(gathered)
        keep = []
        for r in range(world):
            a, b = bdist.shard_range(GB, r, world)
            keep += list(range(r * bmax, r * bmax + (b - a)))
        maps = [mm[keep] for mm in maps]
        # online-eval accumulator (9 error sums + count, bts_main.py:258-260) and the strided eval sharding
        acc = torch.tensor([float(rank + 1)] * 9 + [float(len(bdist.shard_indices(7, rank, world)))])
        bdist.all_reduce_eval_measures(acc)
        assert acc[-1].item() == 7 and acc[0].item() == sum(range(1, world + 1))
        # abs_min over all ranks: all-reduce(min) of the three per-rank LPG scalars
        from types import SimpleNamespace
        lp = lambda v: SimpleNamespace(abs_min=torch.tensor(float(v)))
        fake_dec = SimpleNamespace(lpg8x8=lp(0.5 + rank), lpg4x4=lp(3.0 - rank), lpg2x2=lp(0.25))
        am, _ = bdist.all_reduce_abs_min(fake_dec)
        assert am.tolist() == [0.5, 3.0 - (world - 1), 0.25] and fake_dec.lpg4x4.abs_min.item() == 3.0 - (world - 1)
        if rank == 0:
            ref = run_frames(0, GB)
            for i in range(5):
                assert maps[i].shape == ref[i].shape
                assert torch.equal(maps[i], ref[i]), "gathered map %d != single-process result" % i
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_shard_indices_cover_without_duplicates():
    for total in (1, 5, 16, 697):
        for world in (1, 2, 3, 8):
            got = sorted(i for r in range(world) for i in bdist.shard_indices(total, r, world))
            assert got == list(range(total))
            sizes = [len(bdist.shard_indices(total, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_shard_range_partitions():
    for total in (1, 3, 16, 17, 64):
        for world in (1, 2, 3, 8):
            spans = [bdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_two_rank_broadcast_shard_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_single_process_paths_are_noops():
    m = torch.nn.Linear(3, 2)
    bdist.broadcast_module(m)                                   # not initialised: no-op
    outs = [torch.full((2, 1, 4, 4), float(i)) for i in range(6)]
    g, work = bdist.all_gather_depths(outs, 5)
    assert work is None and tuple(g.shape) == (1, 5, 2, 1, 4, 4)
    maps = bdist.unshard_depths(g)
    assert len(maps) == 5 and all(torch.equal(maps[i], outs[i]) for i in range(5))
