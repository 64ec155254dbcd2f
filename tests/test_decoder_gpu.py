"""GPU parity of the whole decoder hot path (bts_amd.bts.bts.forward, all HIP) against the CPU oracle
and the reference-generated goldens; plus size-independent properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest
import torch

from parity_util import (CONFIGS, OUT_NAMES, build_hip_decoder, check_outputs, hip_run, make_inputs, max_rel,
                         oracle_run, singular_masks)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def decoders():
    assert torch.cuda.is_available()
    return {c: build_hip_decoder(c) for c in CONFIGS}


@pytest.mark.parametrize("cname", ["K", "N"])
def test_decoder_small_vs_oracle_and_golden(golden_dir, decoders, cname):
    g = np.load(os.path.join(golden_dir, "decoder_small.npz"))
    ref_outs, inter = oracle_run(cname, 2, 64, 96, 4321)
    got = hip_run(decoders[cname], cname, 2, 64, 96, 4321)
    rep = check_outputs(got, ref_outs, inter, what=cname + " small")
    print(cname, "small max-rel:", rep)
    # the same comparison against the committed reference outputs
    gold = [torch.from_numpy(g["%s_%s" % (cname, n)]) for n in OUT_NAMES]
    check_outputs(got, gold, inter, what=cname + " small/golden")
    am = [decoders[cname].lpg8x8.abs_min.item(), decoders[cname].lpg4x4.abs_min.item(), decoders[cname].lpg2x2.abs_min.item()]
    np.testing.assert_allclose(am, g["%s_abs_min" % cname], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("cname", ["K", "N"])
def test_decoder_full_size_vs_golden_samples(golden_dir, decoders, cname):
    """B=1 at the BASELINE shapes (352x1216 / 416x544): 4096 sampled pixels per output + range stats."""
    g = np.load(os.path.join(golden_dir, "decoder_full_samples.npz"))
    _, _, _, H, W = CONFIGS[cname]
    got = hip_run(decoders[cname], cname, 1, H, W, 1234)
    for i, n in enumerate(OUT_NAMES):
        flat = got[i].detach().cpu().numpy().reshape(-1)
        idx, val = g["%s_%s_idx" % (cname, n)], g["%s_%s_val" % (cname, n)]
        gv = flat[idx]
        if n == "iconv1":
            assert (np.abs(gv - val) <= 1e-4 + 1e-3 * np.abs(val)).all()
        else:
            # every sample, no allowance: the three LPG maps mask the near-singular pixels exactly -- the golden stores
            # |denominator| of the REFERENCE's own plane equations at each sampled pixel (gen_golden.decoder_full)
            keep = np.ones(idx.shape, dtype=bool)
            if "%s_%s_absden" % (cname, n) in g.files:
                keep = g["%s_%s_absden" % (cname, n)] > 2e-3
                assert keep.mean() > 0.99
            rel = np.abs(gv - val)[keep] / np.maximum(np.abs(val[keep]), 1e-30)
            assert rel.max() <= 1e-4, "%s %s: worst sample off by %g (%d of %d above 1e-4)" % (cname, n, rel.max(), int((rel > 1e-4).sum()), rel.size)
        st = g["%s_%s_stats" % (cname, n)]
        fin = flat[np.isfinite(flat)]
        assert abs(fin.astype(np.float64).mean() - st[2]) <= 1e-3 * max(abs(st[3]), 1e-6) + 1e-3 * abs(st[2])


@pytest.mark.parametrize("cname", ["K", "N"])
def test_decoder_full_size_b16_properties(decoders, cname):
    """BASELINE configs 2 and 3 (B=16 at 352x1216 DenseNet161 plan / 416x544 ResNeXt101 plan) are too slow for the
    CPU oracle in a test; check size-independent properties:
    (1) frames are independent: batch of 16 == 16 batches of 1 (same kernels, bit-exact);
    (2) final_depth is linear in focal for kitti, independent of it for nyu (bts.py:290-291);
    (3) outputs are finite / in range where the maths guarantees it."""
    B = 16
    _, _, _, H, W = CONFIGS[cname]
    dec = decoders[cname]
    feats, focal = make_inputs(cname, B, H, W, 77)
    feats_d = [None] + [f.cuda() for f in feats[1:]]
    with torch.no_grad():
        full = [o.clone() for o in dec(feats_d, focal.cuda())]
        for b in (0, 7, 15):
            one = dec([None] + [f[b:b + 1].contiguous() for f in feats_d[1:]], focal[b:b + 1].cuda())
            for i in range(6):
                assert torch.equal(one[i][0], full[i][b]), "frame %d output %s depends on its batch" % (b, OUT_NAMES[i])
        full2 = dec(feats_d, (focal * 2).cuda())
    torch.testing.assert_close(full2[4], full[4] * (2 if CONFIGS[cname][2] == "kitti" else 1), rtol=1e-6, atol=0)
    for i in (3, 4, 5):
        assert torch.isfinite(full[i]).all()
    assert (full[3] > 0).all() and (full[3] < 1).all()


def test_decoder_rejects_cpu_tensors_in_both_modes(decoders):
    """No CPU fallback: host tensors raise in eval and in train() mode; train() mode on the GPU builds the autograd
    graph of bts_amd/train.py (checked in depth by tests/test_train_gpu.py)."""
    import copy
    dec = decoders["K"]
    feats, focal = make_inputs("K", 1, 64, 96, 1)
    with pytest.raises(RuntimeError):
        dec(feats, focal)                         # CPU tensors: no fallback
    tr = copy.deepcopy(dec).train()               # a copy: train() mode updates the BN running buffers
    with pytest.raises(RuntimeError):
        tr(feats, focal)
    outs = tr([None] + [f.cuda() for f in feats[1:]], focal.cuda())
    assert len(outs) == 6 and outs[4].requires_grad and tuple(outs[4].shape) == (1, 1, 64, 96)


def test_decoder_validates_tap_shapes(decoders):
    """Mismatched encoder taps are refused on the host (no kernel ever sees inconsistent sizes)."""
    from bts_amd._lib import BtsHipError
    dec = decoders["K"]
    feats, focal = make_inputs("K", 1, 64, 96, 3)
    g = [None] + [f.cuda() for f in feats[1:]]
    bad = list(g)
    bad[3] = g[3][:, :, :-1]                                   # wrong height at H/8
    with pytest.raises(BtsHipError):
        dec(bad, focal.cuda())
    bad = list(g)
    bad[2] = g[2][:, :-1]                                      # wrong channel count
    with pytest.raises(BtsHipError):
        dec(bad, focal.cuda())
    with pytest.raises(BtsHipError):
        dec(g[:5], focal.cuda())                               # short list
    with pytest.raises(BtsHipError):
        dec(g, torch.ones(3, device="cuda"))                   # focal per frame
    out = dec(g, focal.cuda())                                 # and the good call still works afterwards
    assert out[4].shape == (1, 1, 64, 96)


@pytest.mark.parametrize("B,H,W", [(1, 32, 32), (3, 32, 96), (5, 96, 32)])
def test_decoder_odd_batches_and_minimum_sizes(decoders, B, H, W):
    """Smallest legal image (one 1/32-resolution pixel), odd batch sizes, non-square maps: every tile is ragged."""
    ref_outs, inter = oracle_run("K", B, H, W, 90 + B)
    got = hip_run(decoders["K"], "K", B, H, W, 90 + B)
    check_outputs(got, ref_outs, inter, what="K %dx%dx%d" % (B, H, W))


def test_module_level_forwards(decoders):
    """The reference's per-module forwards (NCHW in/out) also run on HIP."""
    from oracle import bts_oracle as O
    from bts_amd import synth
    dec = decoders["K"]
    state = O.state_from_numpy(synth.decoder_state(synth.ENCODER_CHANNELS["densenet161_bts"], 512, 0))
    rng = np.random.Generator(np.random.PCG64(9))
    x = torch.from_numpy(rng.standard_normal(size=(2, 128, 6, 10), dtype=np.float32))
    with torch.no_grad():
        y = dec.reduc8x8(x.cuda()).cpu()
        ref = O.reduction_forward(x, O._reduc_weights(state, "reduc8x8"), 80.0, False)
        torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-5)
        pe = torch.cat([torch.nn.functional.normalize(ref[:, :3], 2, 1), ref[:, 3:]], 1)
        d = dec.lpg8x8(pe.cuda(), None).cpu()
        dref, am = O.lpg_forward(pe, 8)
        assert torch.equal(d, dref) and dec.lpg8x8.abs_min.item() == am.item()
        x6 = torch.from_numpy(rng.standard_normal(size=(1, 576, 6, 10), dtype=np.float32))
        torch.testing.assert_close(dec.daspp_6(x6.cuda()).cpu(), O.atrous_forward(x6, state, "daspp_6", 6, True),
                                   rtol=1e-4, atol=2e-5)
        xu = torch.from_numpy(rng.standard_normal(size=(1, 128, 5, 7), dtype=np.float32))
        torch.testing.assert_close(dec.upconv3(xu.cuda()).cpu(), O.upconv_forward(xu, state["upconv3.conv.weight"]),
                                   rtol=1e-4, atol=2e-5)


EMU_CONV_CASES = [
    # B, cin, cout, h, w, k, dil, stride, up/subpixel, nchw_out
    (2, 256, 128, 11, 19, 3, 6, 1, "plain", False),       # ASPP dilated 3x3
    (2, 576, 256, 11, 19, 1, 1, 1, "plain", False),       # ASPP 1x1
    (1, 228, 128, 12, 20, 3, 1, 1, "plain", False),       # odd channel count (general loader)
    (2, 36, 32, 16, 24, 3, 1, 1, "plain", True),          # conv1: NCHW output
    (2, 64, 32, 8, 12, 3, 1, 1, "subpixel", False),       # sub-pixel upconv
    (2, 4, 96, 32, 48, 7, 1, 2, "plain", False),          # stem: stride 2, 7x7
    (1, 2208, 512, 2, 3, 3, 1, 1, "subpixel", False),     # upconv5: K = 8832 per class
]


@pytest.mark.parametrize("case", EMU_CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16x3_emulation_is_fp32_accurate(case):
    """bts_conv_desc.precision = 1 (fp32 emulated on the bf16 matrix cores: three-way operand split, six products)
    against fp64 torch on the CPU, beside the fp32-MFMA mode on the same inputs: both must sit at fp32 rounding level,
    and the emulation may not be more than 3x further from exact than the fp32 MFMA chain."""
    import torch.nn.functional as F
    from bts_amd import ops
    B, cin, cout, h, w, k, dil, stride, mode, nchw = case
    gen = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn(B, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, k, k, generator=gen) / np.sqrt(cin * k * k)
    pad = dil * (k // 2)
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if mode == "subpixel" else x.double()
    ref = F.elu(F.conv2d(xin, wt.double(), stride=stride, padding=pad, dilation=dil))
    H, W = ref.shape[2:]
    c4 = (cin + 3) // 4 * 4
    x2d = torch.zeros(B * h * w, c4, device="cuda")
    x2d[:, :cin] = x.cuda().permute(0, 2, 3, 1).reshape(B * h * w, cin)
    if mode == "subpixel":
        wp, _, _ = ops.pack_upconv_subpixel(wt.cuda(), c_in_ld=c4)
    else:
        wp, _, _ = ops.pack_conv_weight(wt.cuda(), c_in_ld=c4)
    errs = {}
    for prec in ("fp32", "bf16x3"):
        with ops.launch_config(precision=prec):
            if nchw:
                y = torch.empty(B, cout, H, W, device="cuda")
                ops.conv_forward(x2d, B, h, w, wp, cout, k, dil=dil, stride=stride, pad=pad, c_in_ld=c4, act=ops.ACT_ELU,
                                 y_nchw=y, c_in_real=cin)
                got = y.cpu().double()
            else:
                y = torch.empty(B * H * W, cout, device="cuda")
                ops.conv_forward(x2d, B, h, w, wp, cout, 3 if mode == "subpixel" else k, dil=dil, stride=stride,
                                 pad=pad if mode != "subpixel" else None, up=2 if mode == "subpixel" else 1, c_in_ld=c4,
                                 act=ops.ACT_ELU, y2d=y, subpixel=(mode == "subpixel"), c_in_real=cin)
                got = y.view(B, H, W, cout).permute(0, 3, 1, 2).cpu().double()
        errs[prec] = (got - ref).abs().max().item() / ref.abs().max().item()
    print(case, errs)
    assert errs["fp32"] <= 1e-5 and errs["bf16x3"] <= 1e-5, errs
    assert errs["bf16x3"] <= 3.0 * errs["fp32"] + 2e-7, errs


@pytest.mark.parametrize("cname", ["K", "N"])
def test_decoder_bf16x3_mode_vs_oracle(decoders, cname):
    """The whole decoder with every convolution in the bf16x3-emulated mode: same parity bar as the default path."""
    from bts_amd import ops
    ref_outs, inter = oracle_run(cname, 2, 64, 96, 4321)
    dec = decoders[cname]
    dec.conv_precision = "bf16x3"
    try:
        got = hip_run(dec, cname, 2, 64, 96, 4321)
    finally:
        dec.conv_precision = "fp32"
    rep = check_outputs(got, ref_outs, inter, what=cname + " small / bf16x3")
    print(cname, "bf16x3 max-rel:", rep)


def test_conv_bf16x3_wide_dynamic_range():
    """The split keeps fp32's exponent range (bf16 pieces share it), so activations spanning ten decades and weights
    spanning six lose nothing against the fp32-MFMA mode: per-OUTPUT relative error on outputs that are not
    cancellation-dominated, both modes against fp64."""
    import torch.nn.functional as F
    from bts_amd import ops
    B, cin, cout, h, w = 1, 256, 64, 12, 16
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(B, cin, h, w, generator=gen) * torch.pow(10.0, torch.empty(B, cin, 1, 1).uniform_(-6, 4, generator=gen))
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * torch.pow(10.0, torch.empty(cout, 1, 1, 1).uniform_(-4, 2, generator=gen))
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), padding=1)            # sum of |terms|: the scale rounding acts on
    x2d = x.cuda().permute(0, 2, 3, 1).reshape(B * h * w, cin).contiguous()
    wp, _, _ = ops.pack_conv_weight(wt.cuda())
    worst = {}
    for prec in ("fp32", "bf16x3"):
        with ops.launch_config(precision=prec):
            y = torch.empty(B * h * w, cout, device="cuda")
            ops.conv_forward(x2d, B, h, w, wp, cout, 3, y2d=y)
        got = y.view(B, h, w, cout).permute(0, 3, 1, 2).cpu().double()
        worst[prec] = ((got - ref).abs() / mag).max().item()
    print("error / sum|terms|:", worst)
    assert worst["fp32"] <= 2e-6 and worst["bf16x3"] <= 2e-6, worst
