#!/usr/bin/env python3
"""Batch-1 latency of BtsModel.forward at 352x1216 (the reference's real inference loop is B=1, eager, one .cpu() per
frame: pytorch/bts_test.py:127-147), with the host-vs-GPU split, in three execution modes:
  eager : one ctypes crossing per launch (~120 per frame)
  plan  : BtsModel.use_plans -- ONE bts_plan_run call per frame (bts_amd/plan.py)
  graph : hipGraph replay (bts_amd/graph.py; static outputs)
For each: wall time per frame with a device sync after every frame (what bts_test.py sees), host time to ENQUEUE a
frame (no sync inside the loop), and GPU time per frame (HIP events around a queued batch of frames).
    python scripts/latency_b1.py [--encoder densenet161_bts] [--frames 50]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoder", default="densenet161_bts")
    ap.add_argument("--height", type=int, default=352)
    ap.add_argument("--width", type=int, default=1216)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--fill-frames", type=int, default=0,
                    help="pin BtsModel.fill_frames; 0 = the model's default (by the batch of the call: B <= 2 -> 2, B <= 11 -> 8, else 16)")
    a = ap.parse_args()
    import bench
    from bts_amd import ops, synth
    from bts_amd.graph import GraphedModel
    params = bench.Params(a.encoder, 512, 80.0, "kitti")
    model = bench.build_model(params, torch.device("cuda"), seed=0)
    model.sub_batches = 1 if a.batch == 1 else 4
    model.fill_frames = a.fill_frames or None
    img = torch.from_numpy(synth.image_batch(a.batch, a.height, a.width, 1234)).cuda()
    foc = torch.from_numpy(synth.focal_values(a.batch, "kitti", 1234)).cuda()
    out = {"config": "%s, B=%d, 3x%dx%d fp32, %d frames per measurement, fill_frames=%d"
           % (a.encoder, a.batch, a.height, a.width, a.frames, a.fill_frames)}
    gm = GraphedModel(model)
    with torch.no_grad():
        for mode in ("eager", "plan", "graph"):
            model.use_plans = mode == "plan"
            fwd = gm if mode == "graph" else model
            for _ in range(5):
                fwd(img, foc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.frames):
                o = fwd(img, foc)
                torch.cuda.synchronize()                     # bts_test.py consumes every frame on the host
            wall = (time.perf_counter() - t0) / a.frames
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.frames):
                fwd(img, foc)
            e.record()
            torch.cuda.synchronize()
            gpu = s.elapsed_time(e) / a.frames
            # host cost of enqueuing ONE frame into an empty queue (no back-pressure from the device): best of 10
            host = 1e9
            for _ in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fwd(img, foc)
                host = min(host, time.perf_counter() - t0)
            out[mode] = {"ms_per_frame_synced": round(1e3 * wall, 3), "host_enqueue_ms_per_frame": round(1e3 * host, 3),
                         "gpu_ms_per_frame_queued": round(gpu, 3), "frames_per_s_synced": round(a.batch / wall, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
