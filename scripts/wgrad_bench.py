#!/usr/bin/env python3
"""Per-shape timing of bts_conv_wgrad_f32 on the training configuration's layers (B=4, 352x704, DenseNet161-BTS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bts_amd import ops

SHAPES = [  # name, B, h, w, cin, cout, k, dil, up
    ("b1 1x1", 4, 88, 176, 192, 192, 1, 1, 1), ("b1 3x3", 4, 88, 176, 192, 48, 3, 1, 1),
    ("b2 1x1", 4, 44, 88, 480, 192, 1, 1, 1), ("b2 3x3", 4, 44, 88, 192, 48, 3, 1, 1),
    ("b3 1x1", 4, 22, 44, 1248, 192, 1, 1, 1), ("b3 3x3", 4, 22, 44, 192, 48, 3, 1, 1),
    ("b4 1x1", 4, 11, 22, 1632, 192, 1, 1, 1), ("b4 3x3", 4, 11, 22, 192, 48, 3, 1, 1),
    ("aspp 1x1", 4, 44, 88, 704, 256, 1, 1, 1), ("aspp 3x3 d12", 4, 44, 88, 256, 128, 3, 12, 1),
    ("upconv5", 4, 11, 22, 2208, 512, 3, 1, 2), ("conv5", 4, 22, 44, 896, 512, 3, 1, 1),
    ("upconv3", 4, 44, 88, 128, 128, 3, 1, 2), ("conv3", 4, 88, 176, 228, 128, 3, 1, 1),
    ("conv2", 4, 176, 352, 164, 64, 3, 1, 1), ("upconv1", 4, 176, 352, 64, 32, 3, 1, 2),
    ("conv1", 4, 352, 704, 36, 32, 3, 1, 1), ("reduc 32->16", 4, 352, 704, 32, 16, 1, 1, 1),
    ("reduc 8->4", 4, 352, 704, 8, 4, 1, 1, 1), ("reduc 128->64", 4, 88, 176, 128, 64, 1, 1, 1),
]

def main():
    dev = torch.device("cuda:0")
    ws = torch.empty(48 << 20, device=dev)
    for name, B, h, w, cin, cout, k, dil, up in SHAPES:
        H, W = h * up, w * up
        x = torch.randn(B * h * w, cin, device=dev)
        dy = torch.randn(B * H * W, cout, device=dev)
        for _ in range(3):
            ops.conv_wgrad(x, B, h, w, cin, dy, cout, k, dil=dil, up=up, ws=ws)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        s.record()
        for _ in range(n):
            ops.conv_wgrad(x, B, h, w, cin, dy, cout, k, dil=dil, up=up, ws=ws)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / n * 1e3
        fl = 2.0 * B * H * W * cout * cin * k * k
        by = 4.0 * (B * h * w * cin + B * H * W * cout)
        print("%-14s px %7d  M=%4d N=%6d  %8.1f us  %6.1f TF/s  %6.0f GB/s" % (name, B * H * W, cout, cin * k * k, us, fl / us / 1e6, by / us / 1e3), flush=True)

if __name__ == "__main__":
    main()
