#!/usr/bin/env python3
"""Per-layer microbenchmark of the fp32-MFMA conv kernel on the decoder's (and encoder's) GEMM shapes.
    python scripts/conv_bench.py [--batch 16] [--reps 5]
Env BTS_CONV_BM=64|128 forces the row tile (A/B of the tile heuristic)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bts_amd import ops


def layers(B, H=352, W=1216):
    def hw(s):
        return H // s, W // s
    L = [("upconv5", hw(32), 2208, 512, 3, 1, 2), ("conv5", hw(16), 896, 512, 3, 1, 1),
         ("upconv4", hw(16), 512, 256, 3, 1, 2), ("conv4", hw(8), 448, 256, 3, 1, 1)]
    for c, d in ((256, 3), (576, 6), (704, 12), (832, 18), (960, 24)):
        L += [("daspp%d.a" % d, hw(8), c, 256, 1, 1, 1), ("daspp%d.b" % d, hw(8), 256, 128, 3, d, 1)]
    L += [("daspp_conv", hw(8), 896, 128, 3, 1, 1), ("upconv3", hw(8), 128, 128, 3, 1, 2),
          ("conv3", hw(4), 228, 128, 3, 1, 1), ("upconv2", hw(4), 128, 64, 3, 1, 2), ("conv2", hw(2), 164, 64, 3, 1, 1),
          ("upconv1", hw(2), 64, 32, 3, 1, 2), ("conv1", hw(1), 36, 32, 3, 1, 1)]
    return L


def enc_layers(B, H=352, W=1216):
    """DenseNet161 dense layers (first / middle / last of each block) and transitions: (name, (h,w), cin, cout, k, dil, up)."""
    L = []
    for bi, (s, c0, n) in enumerate(((4, 96, 6), (8, 192, 12), (16, 384, 36), (32, 1056, 24))):
        h, w = H // s, W // s
        for li in sorted({0, n // 2, n - 1}):
            cin = c0 + 48 * li
            L += [("b%d.l%d.1x1" % (bi + 1, li), (h, w), cin, 192, 1, 1, 1), ("b%d.l%d.3x3" % (bi + 1, li), (h, w), 192, 48, 3, 1, 1)]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--set", default="dec", choices=["dec", "enc", "all"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"])
    ap.add_argument("--fill-frames", type=int, default=16)
    a = ap.parse_args()
    B = a.batch
    tot_ms = tot_fl = 0.0
    todo = (layers(B) if a.set in ("dec", "all") else []) + (enc_layers(B) if a.set in ("enc", "all") else [])
    for name, (h, w), cin, cout, k, dil, up in todo:
        if a.only and a.only not in name:
            continue
        x = torch.randn(B * h * w, cin, device="cuda")
        wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        sub = up == 2 and k == 3
        wp, cop, kp = ops.pack_upconv_subpixel(wt) if sub else ops.pack_conv_weight(wt)
        y = torch.empty(B * h * up * w * up, cout, device="cuda")
        pre = (torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1) if name.endswith(".1x1") else None   # DenseNet bottleneck: norm1 + relu on the way in
        run = lambda: ops.conv_forward(x, B, h, w, wp, cout, k, dil=dil, up=up, act=ops.ACT_ELU, y2d=y, subpixel=sub, pre=pre, pre_relu=pre is not None)
        with ops.launch_config(fill_frames=a.fill_frames, precision=a.precision):
            run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(a.reps):
                s.record(); run(); e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e))
        M = B * h * up * w * up
        fl = 2.0 * M * cout * cin * (4 if sub else k * k)       # EXECUTED flops (sub-pixel upconv: 4 taps per output)
        tot_ms += best
        tot_fl += fl
        print("%-11s M=%8d K=%6d N=%4d  %8.1f us  %6.1f TF" % (name, M, cin * k * k, cout, best * 1e3, fl / best / 1e9), flush=True)
    print("total %.3f ms  %.1f TF" % (tot_ms, tot_fl / tot_ms / 1e9))


if __name__ == "__main__":
    main()
