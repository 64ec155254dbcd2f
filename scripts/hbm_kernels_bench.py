#!/usr/bin/env python3
"""Bandwidth of the HBM-bound hot-path kernels (LPG, reduction stacks) at the bench shape (B=16) and at a size
large enough to amortise launch latency (a B=16 LPG writes 27 MB = 3.4 us at 8 TB/s, the same order as a launch).
    python scripts/hbm_kernels_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bts_amd import ops, synth

PEAK = 8000.0


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best


def lpg(B, k, H=352, W=1216, ds=True):
    h, w = H // k, W // k
    plane4 = torch.rand(B * h * w, 4, device="cuda") + 0.5
    out = torch.empty(B, 1, H, W, device="cuda")
    am = torch.zeros(1, device="cuda")
    f = {8: 4, 4: 2, 2: 1}[k]
    dsb = torch.zeros(B * (H // f) * (W // f), 4, device="cuda") if (ds and f > 1) else None
    fn = lambda: ops.lpg_fused_forward(plane4, B, h, w, k, 80.0, False, out, ds_out=dsb[:, 0] if dsb is not None else None,
                                       ds_factor=f, ds_pix_stride=4, abs_min=am)
    ms = timeit(fn)
    nbytes = 4.0 * (plane4.numel() + out.numel() + (dsb.shape[0] if dsb is not None else 0))
    print("lpg k=%d B=%3d %s: %8.1f us  %7.1f GB/s (%4.1f %% of 8 TB/s)" % (k, B, "with ds " if dsb is not None else "no ds   ", ms * 1e3, nbytes / ms / 1e6, nbytes / ms / 1e6 / PEAK * 100))


def reduc(B, name, H=352, W=1216):
    cin, cfirst, fin, s = {"8x8": (128, 128, False, 8), "4x4": (128, 64, False, 4), "2x2": (64, 32, False, 2), "1x1": (32, 16, True, 1)}[name]
    npix = B * (H // s) * (W // s)
    x = torch.randn(npix, cin, device="cuda")
    chain = synth.reduc_chain_channels(cin, cfirst, fin)
    ws = [torch.randn(chain[i + 1], chain[i], 1, 1, device="cuda") * 0.1 for i in range(len(chain) - 1)]
    frag = ops.pack_reduc_weights(ws)
    out = torch.empty(npix * (1 if fin else 4), device="cuda")
    ms = timeit(lambda: ops.reduc_forward_nhwc(x, cin, cfirst, frag, 80.0, fin, True, out))
    nbytes = 4.0 * (x.numel() + out.numel())
    macs = sum(chain[i] * chain[i + 1] for i in range(len(chain) - 1))
    print("reduc%s B=%3d: %8.1f us  %7.1f GB/s (%4.1f %% of 8 TB/s)  %6.1f TFLOP/s" % (name, B, ms * 1e3, nbytes / ms / 1e6, nbytes / ms / 1e6 / PEAK * 100, 2.0 * npix * macs / ms / 1e9))


def reduc_lpg(B, name, H=352, W=1216):
    """The fused launch of the decoder pipeline: reduction chain -> normalize -> LPG -> /max_depth (+ downsampled plane, abs_min)."""
    cin, cfirst, k = {"8x8": (128, 128, 8), "4x4": (128, 64, 4), "2x2": (64, 32, 2)}[name]
    h, w = H // k, W // k
    npix = B * h * w
    x = torch.randn(npix, cin, device="cuda")
    chain = synth.reduc_chain_channels(cin, cfirst, False)
    ws = [torch.randn(chain[i + 1], chain[i], 1, 1, device="cuda") * 0.1 for i in range(len(chain) - 1)]
    frag = ops.pack_reduc_weights(ws)
    depth = torch.empty(B, 1, H, W, device="cuda")
    ds = torch.empty(npix * 4, device="cuda") if k > 2 else None
    am = torch.zeros((), device="cuda")
    ms = timeit(lambda: ops.reduc_lpg_forward(x, B, h, w, cin, cfirst, frag, 80.0, k, depth, ds_out=ds, abs_min=am))
    nbytes = 4.0 * (x.numel() + depth.numel() + (ds.numel() if ds is not None else 0))
    macs = sum(chain[i] * chain[i + 1] for i in range(len(chain) - 1))
    print("reduc_lpg%s B=%3d: %8.1f us  %7.1f GB/s (%4.1f %% of 8 TB/s)  %6.1f TFLOP/s" % (name, B, ms * 1e3, nbytes / ms / 1e6, nbytes / ms / 1e6 / PEAK * 100, 2.0 * npix * macs / ms / 1e9))


if "--reduc-only" in sys.argv:
    for n in ("8x8", "4x4", "2x2"):
        reduc_lpg(16, n)
    for n in ("8x8", "4x4", "2x2", "1x1"):
        reduc(16, n)
    sys.exit(0)

# calibration: what this device sustains for pure streaming writes / copies (torch elementwise kernels, 1 GiB)
buf = torch.empty(256 * 1024 * 1024, device="cuda")
src = torch.randn_like(buf)
ms = timeit(lambda: buf.fill_(1.0), reps=5)
print("calibration write-only (fill 1 GiB):      %7.1f GB/s" % (buf.numel() * 4 / ms / 1e6))
ms = timeit(lambda: buf.copy_(src), reps=5)
print("calibration copy (1 GiB read + 1 GiB write): %7.1f GB/s" % (2 * buf.numel() * 4 / ms / 1e6))
ms = timeit(lambda: torch.sum(src), reps=5)
print("calibration read-only (sum 1 GiB):        %7.1f GB/s" % (buf.numel() * 4 / ms / 1e6))
del buf, src

for B in (16, 128):
    for k in (8, 4, 2):
        lpg(B, k)
    lpg(B, 8, ds=False)
    for n in ("8x8", "4x4", "2x2", "1x1"):
        reduc(B, n)
    for n in ("8x8", "4x4", "2x2"):
        reduc_lpg(B, n)
