#!/usr/bin/env python3
"""Training-step timing of BtsModel on one MI355X (the reference's training
configuration: DenseNet161, batch 4 per GPU, 352x704 crops, bts_main.py / arguments_train_eigen.txt).

Prints step time (forward + silog loss + backward + AdamW update) and, with --trace, the per-kernel split of the
HIP convolutions (forward / dgrad / wgrad) with their algorithmic TFLOP/s.  A tool for the perf loop; the headline
bench.py metric stays the inference frames/s.
"""
import argparse
import json
import sys
import time
from collections import namedtuple

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bts_amd import bts as M, ops, synth, trainer  # noqa: E402

Params = namedtuple("Params", "encoder bts_size max_depth dataset")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=352)
    ap.add_argument("--width", type=int, default=704)
    ap.add_argument("--encoder", default="densenet161_bts")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--decoder-only", action="store_true")
    ap.add_argument("--no-freeze", action="store_true", help="train every encoder parameter (the reference freezes some)")
    a = ap.parse_args()
    # BASELINE config 5 (B=32 over 8 GPUs = 4 per GPU, DDP): launch with
    #   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_bench.py
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("BTS_BENCH_FORCE_DIST") == "1"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    saved_stdout = os.dup(1)                 # RCCL prints a banner on stdout: keep the JSON line clean
    os.dup2(2, 1)
    torch.manual_seed(0)
    params = Params(a.encoder, 512, 80.0, "kitti")
    model = M.BtsModel(params).train().to(dev)
    loss_fn = M.silog_loss(0.85)
    if not a.no_freeze:
        trainer.set_misc(model, a.encoder)          # the reference freezes the stem conv and the encoder norm affines
    opt = trainer.make_optimizer(model, 1e-4, 1e-2, 1e-3)
    core = model
    if use_dist:
        model = trainer.wrap_ddp(model, dev)   # gradient all-reduce over RCCL, overlapped with backward by DDP's buckets
    B, H, W = a.batch, a.height, a.width
    x = torch.from_numpy(synth.image_batch(B, H, W, 1)).to(dev)
    focal = torch.from_numpy(synth.focal_values(B, "kitti", 1)).to(dev)
    gt, mask = synth.train_targets(B, H, W, 80.0, 2)
    gt, mask = torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev)
    feats = None
    if a.decoder_only:
        fs = synth.encoder_features(synth.ENCODER_CHANNELS[a.encoder], B, H, W, seed=3)
        feats = [None] + [torch.from_numpy(f).to(dev).requires_grad_(True) for f in fs[1:]]

    def step():
        opt.zero_grad(set_to_none=True)
        outs = core.decoder(feats, focal) if a.decoder_only else model(x, focal)
        loss = loss_fn(outs[4], gt, mask)
        loss.backward()
        opt.step()
        return loss

    for i in range(a.warmup):
        t0 = time.time()
        l = step()
        torch.cuda.synchronize()
        print("warmup %d: %.1f ms loss %.4f" % (i, (time.time() - t0) * 1e3, l.item()), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.time()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 1e3 / a.steps
    if use_dist:
        dist.barrier()
        tmax = torch.tensor([ms], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ms = tmax.item()
    res = dict(metric="training step ms (fwd+loss+bwd+AdamW)", ms_per_step=ms, frames_per_s=world * B / ms * 1e3,
               n_gpus=world, ddp=use_dist,
               config=dict(encoder=a.encoder, batch_per_gpu=B, height=H, width=W, decoder_only=a.decoder_only),
               peak_mem_gb=torch.cuda.max_memory_allocated() / 2**30)
    if a.trace:
        tr = ops.KernelTrace()
        ops.set_trace(tr)
        step()
        ops.set_trace(None)
        summ = tr.summary()
        rows = {}
        for kern, d in summ.items():
            for tag, v in d["tags"].items():
                kind = tag.rsplit(".", 1)[-1] if "." in tag else kern
                r = rows.setdefault(kind, dict(ms=0.0, flops=0.0, launches=0))
                r["ms"] += v["ms"]
                r["flops"] += v["flops"]
                r["launches"] += v["launches"]
        for k, r in rows.items():
            r["tflops"] = r["flops"] / max(r["ms"], 1e-9) / 1e9
        res["kernels"] = rows
        top = []
        for kern, d in summ.items():
            for tag, v in d["tags"].items():
                top.append((v["ms"], kern, tag, v["launches"], v["flops"] / max(v["ms"], 1e-9) / 1e9))
        top.sort(reverse=True)
        for ms_, kern, tag, n, tf in top[:25]:
            print("%8.2f ms  %-34s %-22s x%-4d %6.1f TF/s" % (ms_, kern, tag, n, tf), file=sys.stderr)
    if rank == 0:
        os.write(saved_stdout, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
