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
        # uneven shards (2 + 1 frames): the library pads to ceil(G / world) frames for the fixed-size
        # all_gather_into_tensor and trims the pad frames again -- DataParallel.scatter / the reference's
        # no-evenly-divisible sampler semantics (distributed_sampler_no_evenly_divisible.py:62), no duplicated frame
        gathered, work = bdist.all_gather_depths(outs, 5, async_op=True, global_batch=GB)
        work.wait()
        assert tuple(gathered.shape[:3]) == (world, 5, 2)
        maps = bdist.unshard_depths(gathered, GB)
        # persistent buffers: a DepthGather allocates once; the model writes its maps straight into the send buffer
        # (BtsModel.output_buffers = dg.outputs(slot)), two slots alternate so a collective may still read slot i while
        # step i+1 fills the other one
        dg = bdist.DepthGather(5, 2, H, W, "cpu", slots=2)
        ptrs = [(p.data_ptr(), g.data_ptr()) for p, g in zip(dg.packed, dg.gathered)]
        for stepno in range(4):
            sl = stepno % 2
            views = dg.outputs(sl, hi - lo)
            assert all(v.is_contiguous() and tuple(v.shape) == (hi - lo, 1, H, W) for v in views)
            for v, o in zip(views, outs):
                v.copy_(o + float(stepno))                   # stands for the kernels' stores
            w = dg.gather(sl, async_op=True)
            w.wait()
            got = dg.maps(sl, GB)
            for i in range(5):
                assert torch.equal(got[i], maps[i] + float(stepno)), "slot %d, step %d" % (sl, stepno)
        assert ptrs == [(p.data_ptr(), g.data_ptr()) for p, g in zip(dg.packed, dg.gathered)], "buffers were reallocated"
        dg.pack(0, dg.outputs(0, hi - lo))                   # already in place: no copy, no error
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
        # a NaN on ONE rank must come out as NaN on EVERY rank (bts.py:167's min() is NaN as soon as one denominator is;
        # ReduceOp.MIN's own NaN behaviour is the backend's business): it travels as -1
        nan_dec = SimpleNamespace(lpg8x8=lp(float("nan") if rank == world - 1 else 0.5), lpg4x4=lp(1.0 + rank), lpg2x2=lp(float("nan")))
        am, _ = bdist.all_reduce_abs_min(nan_dec)
        assert torch.isnan(am[0]) and am[1].item() == 1.0 and torch.isnan(am[2])
        assert torch.isnan(nan_dec.lpg8x8.abs_min) and nan_dec.lpg4x4.abs_min.item() == 1.0
        am, w = bdist.all_reduce_abs_min(SimpleNamespace(lpg8x8=lp(float("nan") if rank == 0 else 2.0), lpg4x4=lp(4.0), lpg2x2=lp(5.0)),
                                         async_op=True)
        w.wait()
        am = bdist.decode_abs_min(am)
        assert torch.isnan(am[0]) and am[1:].tolist() == [4.0, 5.0]
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
    g2, _ = bdist.all_gather_depths(outs, 5)                    # persistent: the second slot, then the first again
    g3, _ = bdist.all_gather_depths(outs, 5)
    assert g2.data_ptr() != g.data_ptr() and g3.data_ptr() == g.data_ptr()
    from types import SimpleNamespace
    dec = SimpleNamespace(**{n: SimpleNamespace(abs_min=torch.tensor(v)) for n, v in
                             (("lpg8x8", float("nan")), ("lpg4x4", 0.5), ("lpg2x2", 2.0))})
    am, _ = bdist.all_reduce_abs_min(dec)
    assert torch.isnan(am[0]) and am[1:].tolist() == [0.5, 2.0] and torch.isnan(dec.lpg8x8.abs_min)
