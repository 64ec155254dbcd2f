#!/usr/bin/env python3
"""Which host lines launch the small ATen / runtime kernels (fills, copies) of one eval forward?
    python scripts/small_launches_probe.py [--batch 16] [--plans]
torch.profiler with stacks around ONE eager forward after warm-up: prints every non-bts kernel with its count and the
innermost bts_amd / bench frame that caused it."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--plans", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = bench.build_model(bench.Params("densenet161_bts", 512, 80.0, "kitti"), dev)
    model.use_plans = a.plans
    x = torch.randn(a.batch, 3, 352, 1216, device=dev)
    focal = torch.full((a.batch,), 721.5377, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model(x, focal)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            model(x, focal)
            torch.cuda.synchronize()
    by = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.stack:
            frames = [f for f in ev.stack if "bts_amd" in f or "bench.py" in f]
            by[(ev.name, frames[0] if frames else ev.stack[0])] += 1
    for (name, where), n in sorted(by.items(), key=lambda kv: -kv[1])[:60]:
        print("%4d  %-28s %s" % (n, name, where))
    kern = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            kern[ev.name[:70]] += 1
    print("---- device kernels")
    for name, n in kern.most_common(40):
        print("%4d  %s" % (n, name))


if __name__ == "__main__":
    main()
