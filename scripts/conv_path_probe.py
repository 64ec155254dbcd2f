#!/usr/bin/env python3
"""Fast (tap-uniform K-step, c_in % 32 == 0) vs general loader of the conv kernel on neighbouring channel counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bts_amd import ops

def run(name, B, h, w, cin, cout, k, dil):
    x = torch.randn(B * h * w, cin, device="cuda")
    wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wp, cop, kp = ops.pack_conv_weight(wt)
    y = torch.empty(B * h * w, cout, device="cuda")
    f = lambda: ops.conv_forward(x, B, h, w, wp, cout, k, dil=dil, act=ops.ACT_ELU, y2d=y)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record(); f(); e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e))
    fl = 2.0 * B * h * w * cout * cin * k * k
    print("%-22s cin %4d (%s)  %8.1f us  %6.1f TF" % (name, cin, "fast" if cin % 32 == 0 else "general", best * 1e3, fl / best / 1e9), flush=True)

for cin in (160, 164, 192):
    run("conv2 3x3 N=64 @1/2", 16, 176, 608, cin, 64, 3, 1)
for cin in (224, 228, 256):
    run("conv3 3x3 N=128 @1/4", 16, 88, 304, cin, 128, 3, 1)
for cin in (32, 36, 64):
    run("conv1 3x3 N=32 @1/1", 16, 352, 1216, cin, 32, 3, 1)
for cin in (480, 528, 1008, 1056):
    run("dense 1x1 N=192 @1/8", 16, 44, 152, cin, 192, 1, 1)
