"""Round-3 GPU tests: the launch declaration (fill_frames, precision) as a per-model attribute instead of a process
global, and the kernels / host paths added this round."""
import threading

import numpy as np
import pytest
import torch

from bts_amd import synth
from parity_util import Params, t

from conftest import fp32_only

pytestmark = pytest.mark.gpu


def _model(enc="densenet161_bts", dataset="kitti", seed=3):
    from bts_amd import bts as M
    torch.manual_seed(seed)
    m = M.BtsModel(Params(enc, 512, 80.0 if dataset == "kitti" else 10.0, dataset))
    sd = {k: (torch.tensor(v) if np.ndim(v) == 0 else t(v))
          for k, v in synth.decoder_state(synth.ENCODER_CHANNELS[enc], 512, 0).items()}
    m.decoder.load_state_dict(sd, strict=True)
    return m.eval()


def _aspp_kernels(m, img, foc):
    from bts_amd import ops
    tr = ops.KernelTrace()
    ops.set_trace(tr)
    try:
        outs = [o.clone() for o in m(img, foc)]
    finally:
        ops.set_trace(None)
    return outs, sorted({r[0] for r in tr.records if r[1].startswith("aspp")})


def test_two_models_keep_their_own_launch_declaration():
    """Two models with different ``fill_frames`` in ONE process, called alternately and from two threads at once: each
    keeps the bits of its own setting (the declaration is a per-call thread-local scope, not a process global), the
    two settings really differ, and a model left at the default (None) follows the batch of the call: B = 1 gets the
    single-frame latency setting (the ASPP convolutions split K), B = 4 the library default (they do not)."""
    a = _model().cuda()
    b = _model().cuda()
    b.load_state_dict(a.state_dict())
    a.sub_batches = b.sub_batches = 1
    a.fill_frames, b.fill_frames = 8, 2
    H, W = 352, 1216
    img = t(synth.image_batch(1, H, W, 31)).cuda()
    foc = t(synth.focal_values(1, "kitti", 31)).cuda()
    with torch.no_grad():
        ra, ka = _aspp_kernels(a, img, foc)
        rb, kb = _aspp_kernels(b, img, foc)
        assert not any("splitk" in k for k in ka) and any("splitk" in k for k in kb), (ka, kb)
        assert not all(torch.equal(x, y) for x, y in zip(ra, rb)), "the two declarations give the same bits"
        for x, y in zip(ra, rb):
            assert (x - y).abs().max().item() / y.abs().max().item() < 2e-5
        for _ in range(2):                                             # alternating calls: no leakage either way
            assert all(torch.equal(x, y) for x, y in zip(a(img, foc), ra))
            assert all(torch.equal(x, y) for x, y in zip(b(img, foc), rb))
        # two threads, two streams, at the same time
        bad = []

        def work(model, ref):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st), torch.no_grad():
                    for _ in range(3):
                        outs = model(img, foc)
                        st.synchronize()
                        if not all(torch.equal(x, y) for x, y in zip(outs, ref)):
                            bad.append("bits changed under a concurrent model with another declaration")
            except Exception as e:                                      # noqa: BLE001
                bad.append(repr(e))

        th = [threading.Thread(target=work, args=(a, ra)), threading.Thread(target=work, args=(b, rb))]
        [x.start() for x in th]
        [x.join() for x in th]
        assert not bad, bad
        # default (None): by the batch of the call
        c = _model().cuda()
        c.load_state_dict(a.state_dict())
        c.sub_batches = 1
        assert c.fill_frames is None
        r1, k1 = _aspp_kernels(c, img, foc)
        assert all(torch.equal(x, y) for x, y in zip(r1, rb)), "B=1 at the default must be the fill_frames=2 path"
        img4 = t(synth.image_batch(4, H, W, 31)).cuda()
        foc4 = t(synth.focal_values(4, "kitti", 31)).cuda()
        r4, k4 = _aspp_kernels(c, img4, foc4)
        assert not any("splitk" in k for k in k4), k4
        a4 = a(img4, foc4)
        assert all(torch.equal(x, y) for x, y in zip(r4, a4)), "B=4 at the default must be the fill_frames=8 path"


@fp32_only
def test_plans_and_graphs_follow_the_launch_declaration():
    """A recorded plan and a captured graph bake the declaration into every descriptor: after ``conv_precision`` or
    ``fill_frames`` changes on the model, neither may replay the old recording (set_conv_precision did not
    age plans / graphs)."""
    from bts_amd.graph import GraphedModel
    m = _model("densenet121_bts").cuda()
    m.sub_batches = 1
    img = t(synth.image_batch(1, 96, 128, 5)).cuda()
    foc = t(synth.focal_values(1, "kitti", 5)).cuda()
    with torch.no_grad():
        ref = {}
        for prec in ("fp32", "bf16x3"):
            m.conv_precision = prec
            ref[prec] = [o.clone() for o in m(img, foc)]
        assert not all(torch.equal(x, y) for x, y in zip(ref["fp32"], ref["bf16x3"]))
        gm = GraphedModel(m)
        m.use_plans = True
        for prec in ("fp32", "bf16x3", "fp32"):
            m.conv_precision = prec
            for _ in range(2):                                  # record / capture, then replay
                assert all(torch.equal(x, y) for x, y in zip(m(img, foc), ref[prec])), "plan replayed another precision"
            m.use_plans = False
            for _ in range(2):
                assert all(torch.equal(x, y) for x, y in zip(gm(img, foc), ref[prec])), "graph replayed another precision"
            m.use_plans = True
        assert gm.captures == 2 and len(gm._graphs) == 2         # one graph per declaration, the fp32 one reused
        m.use_plans = False
        m.conv_precision = "fp32"
        m.fill_frames = 8
        r8 = [o.clone() for o in m(img, foc)]
        g8 = gm(img, foc)
        assert all(torch.equal(x, y) for x, y in zip(g8, r8)) and gm.captures == 3


# ------------------------------------------------------------------------------- dilated halo tiles (ASPP 3x3, bts.py:73-77)
@fp32_only
@pytest.mark.parametrize("dil,shape", [(3, (2, 44, 152)), (6, (2, 44, 152)), (12, (1, 44, 152)), (3, (1, 46, 150)), (6, (3, 45, 150)),
                                       (12, (2, 47, 160))])
def test_dilated_halo_tile_vs_torch_and_row_tiled(dil, shape):
    """conv_halo_kernel<..., DIL>: the dilated 3x3 of an ASPP branch (BN + ReLU prologue, 256 -> 128, padding = dilation)
    on LDS-staged tiles whose four output rows lie `dil` apart, so that the input patch has six rows whatever the
    dilation.  Against torch in fp64 (zero padding of the POST-prologue tensor on all four borders; maps whose height is
    not a multiple of the 4*dil row group and whose width is not a multiple of 32; several frames) and against the
    row-tiled kernel (same launch with BTS_CONV_HALO_DIL=0 in a child process: both are exact-f32 fmaf chains in a
    different K order, so they agree to rounding, not bit for bit); frames are independent (bit-equal alone and in a
    batch)."""
    import os, subprocess, sys, tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_round2_gpu as T
    if os.environ.get("BTS_CONV_HALO_DIL") != "2":
        # the library default puts only dilation 3 on the dilated tile (6 / 12 measured slower than the tap-skipping row
        # tiles): run this test's body in a child process that enables all three
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_round3_gpu as R; "
                "R.test_dilated_halo_tile_vs_torch_and_row_tiled(%d, %r)" % (
                    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), dil, shape))
        subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, BTS_CONV_HALO_DIL="2"))
        return
    y, ratio, kern = T._dilated_case(dil, shape)
    assert kern == "conv_halo_kernel<128,k3,nhwc,dil>", kern
    B, h, w = shape
    g = torch.Generator().manual_seed(100 + dil)
    x = torch.randn((B * h * w, 256), generator=g)
    wt = torch.randn((128, 256, 3, 3), generator=g) * 0.03
    ps, pb = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    xin = torch.relu(x.double() * ps.double() + pb.double()).reshape(B, h, w, 256).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xin, wt.double(), dilation=dil, padding=dil)
    got = y.double().reshape(B, h, w, 128).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() / ref.abs().max().item() <= 3e-6
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_round2_gpu as T; "
            "y, r, k = T._dilated_case(%d, %r); assert k.startswith('conv_fwd_kernel<'), k; torch.save(y, sys.argv[1])" % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), dil, shape))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "y.pt")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, BTS_CONV_HALO_DIL="0"))
        y_rows = torch.load(out, weights_only=True)
    assert (y_rows - y).abs().max().item() / y.abs().max().item() <= 3e-6
    if B > 1:                                   # the last frame alone: same bits
        from bts_amd import ops
        wp, _, _ = ops.pack_conv_weight(wt.cuda())
        y1 = torch.empty((h * w, 128), device="cuda")
        ops.conv_forward(x[(B - 1) * h * w:].cuda(), 1, h, w, wp, 128, 3, dil=dil, pad=dil, pre=(ps.cuda(), pb.cuda()), pre_relu=True, y2d=y1)
        assert torch.equal(y1.cpu(), y[(B - 1) * h * w:])


@fp32_only
def test_dilated_halo_tile_is_chosen_by_geometry_only():
    """Which dilated 3x3 convolutions take the dilated halo tile by default: dilation 3 with 128 outputs on maps whose
    4*dil row groups and 32-pixel columns fill >= 80 % of the tile grid (the 44x152 KITTI map: yes; the 52x68 NYU map: no);
    every other dilation and single-frame launches that split K stay row-tiled."""
    import ctypes as C
    from bts_amd import _lib, ops
    lib = _lib.load()

    def kind(dil, h, w, fill, cout=128, B=1):
        x = torch.empty((B * h * w, 256), device="cuda")
        wp, _, _ = ops.pack_conv_weight(torch.zeros((cout, 256, 3, 3), device="cuda"))
        y = torch.empty((B * h * w, cout), device="cuda")
        ws = torch.empty(8 * B * h * w * cout, device="cuda")
        tr = ops.KernelTrace()
        ops.set_trace(tr)
        try:
            with ops.launch_config(fill_frames=fill):
                ops.conv_forward(x, B, h, w, wp, cout, 3, dil=dil, pad=dil, y2d=y, splitk_ws=ws)
        finally:
            ops.set_trace(None)
        return sorted(tr.summary())[0]

    assert kind(3, 44, 152, 8) == kind(3, 44, 152, 16) == "conv_halo_kernel<128,k3,nhwc,dil>"
    for d in (6, 12, 18, 24):                                    # 6 / 12: built, measured slower than the tap-skipping row tiles
        assert kind(d, 44, 152, 8).startswith("conv_fwd_kernel<")
    assert kind(3, 52, 68, 8).startswith("conv_fwd_kernel<")
    assert "splitk" in kind(3, 44, 152, 1)                       # single-frame declaration: split-K fills the chip better
    assert kind(3, 44, 152, 8, cout=64).startswith("conv_fwd_kernel<")


# ------------------------------------------------------------------- bf16x3 halo-tile kernel (precision 1, conv_halo_emu.inc)
@pytest.mark.parametrize("cin,cout,shape,mode,pre", [(64, 128, (2, 44, 152), "conv", True), (448, 256, (1, 44, 152), "conv", False),
                                                     (192, 48, (2, 88, 304), "conv", True), (128, 64, (1, 45, 150), "conv", True),
                                                     (128, 128, (2, 44, 152), "subpixel", False), (128, 64, (1, 88, 304), "subpixel", False),
                                                     (32, 128, (3, 11, 93), "conv", True)])
def test_bf16x3_halo_tile_vs_fp64_and_fp32_mode(cin, cout, shape, mode, pre):
    """conv_halo_emu_kernel: stride-1 3x3 / sub-pixel 2x2 convolutions in the fp32-emulated-on-bf16 arithmetic on halo
    tiles -- input patch split into three bf16 planes once per chunk, weights pre-split offline (ops.split_bf16x3) and
    streamed by LDS-DMA.  Against torch in fp64 at the fp32-MFMA mode's own error level (both modes reported), ragged
    right / bottom tiles, prologue + zero padding, ELU epilogue into a strided channel slice, frames independent."""
    import torch.nn.functional as F
    from bts_amd import ops
    B, h, w = shape
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn((B, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(cin * 9.0)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    xin = torch.relu(x.double() * ps.double().view(1, -1, 1, 1) + pb.double().view(1, -1, 1, 1)) if pre else x.double()
    sub = mode == "subpixel"
    if sub:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.elu(F.conv2d(xin, wt.double(), padding=1))
    H, W = ref.shape[2:]
    x2d = x.permute(0, 2, 3, 1).reshape(B * h * w, cin).contiguous().cuda()
    wp = (ops.pack_upconv_subpixel(wt.cuda()) if sub else ops.pack_conv_weight(wt.cuda()))[0]
    ybuf = torch.zeros((B * H * W, cout + 32), device="cuda")
    y = ybuf[:, 16:16 + cout]

    def run(prec, xx, bb, out):
        tr = ops.KernelTrace()
        ops.set_trace(tr)
        try:
            with ops.launch_config(fill_frames=16, precision=prec):
                ops.conv_forward(xx, bb, h, w, wp, cout, 3, up=2 if sub else 1, act=ops.ACT_ELU, y2d=out, subpixel=sub,
                                 pre=(ps.cuda(), pb.cuda()) if pre else None, pre_relu=pre)
        finally:
            ops.set_trace(None)
        return sorted(tr.summary())[0]

    errs = {}
    for prec in ("fp32", "bf16x3"):
        kern = run(prec, x2d, B, y)
        if prec == "bf16x3":
            assert kern == "conv_halo_emu_kernel<%d,k%d>" % (128 if cout >= 128 else 64, 2 if sub else 3), kern
        got = y.reshape(B, H, W, cout).permute(0, 3, 1, 2).cpu().double()
        errs[prec] = (got - ref).abs().max().item() / ref.abs().max().item()
        assert float(ybuf[:, :16].abs().max()) == 0.0 and float(ybuf[:, 16 + cout:].abs().max()) == 0.0     # only the slice is written
    print((cin, cout, shape, mode), errs)
    assert errs["fp32"] <= 1e-5 and errs["bf16x3"] <= 1e-5, errs
    assert errs["bf16x3"] <= 3.0 * errs["fp32"] + 2e-7, errs
    if B > 1:
        full = y.clone()
        y1 = torch.zeros((H * W, cout), device="cuda")
        run("bf16x3", x2d[(B - 1) * h * w:], 1, y1)
        assert torch.equal(y1, full[(B - 1) * H * W:])


def test_split_bf16x3_matches_the_definition():
    """ops.split_bf16x3: h + m + l reproduces w to 2^-24 |w|, every piece is a bf16 (16 significant bits kept as the top
    half of an fp32), and the planes are what a truncation split gives (checked against a NumPy statement)."""
    from bts_amd import ops
    g = torch.Generator().manual_seed(5)
    w = (torch.randn((64, 96), generator=g) * torch.pow(10.0, torch.empty((64, 1)).uniform_(-6, 4, generator=g))).contiguous()
    w[0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.0e-30])
    planes = ops.split_bf16x3(w.cuda()).cpu()
    assert planes.shape == (3, 64, 96) and planes.dtype == torch.int16
    pieces = (planes.to(torch.int32) << 16).view(torch.float32).double()
    rec = pieces.sum(0)
    assert ((rec - w.double()).abs() <= 2.0 ** -24 * w.double().abs() + 1e-45).all()
    wn = w.numpy()
    hb = wn.view(np.uint32) & 0xffff0000
    r1 = wn - hb.view(np.float32)
    mb = r1.view(np.uint32) & 0xffff0000
    lb = (r1 - mb.view(np.float32)).view(np.uint32) & 0xffff0000
    want = np.stack([hb, mb, lb]) >> 16
    assert np.array_equal(planes.numpy().view(np.uint16), want.astype(np.uint16))
    w4 = torch.randn((4, 32, 64), generator=g).cuda()
    assert ops.split_bf16x3(w4).shape == (4, 3, 32, 64)


# ------------------------------------------------------------------------------------ torch operator boundary (TORCH_LIBRARY bts_hip)
def test_torch_ops_validate_and_match_the_ctypes_binding():
    """torch.ops.bts_hip.* (csrc/torch_ops.cpp): the same launches as the ctypes binding -- bit-identical results -- behind
    TORCH_CHECK validation (dtype, shape, contiguity, device), a device guard and the CURRENT stream; autograd on
    bts_hip::lpg equals the oracle's autograd through the reference formulation (bts.py:149-173)."""
    import subprocess, sys, os
    from bts_amd import ops
    from oracle import bts_oracle as O
    tops = ops.torch_ops()
    assert tops is not None
    g = torch.Generator().manual_seed(9)
    pe = (torch.rand((2, 4, 6, 9), generator=g) + 0.5)
    # --- lpg: bit-exact module op, abs_min, autograd
    depth, am = tops.lpg(pe.cuda(), 4)
    ref, ref_am = O.lpg_forward(pe, 4)
    assert torch.equal(depth.cpu(), ref) and am.item() == ref_am.item()
    x = pe.clone().cuda().requires_grad_(True)
    d2, _ = tops.lpg(x, 8)
    wgt = torch.rand(d2.shape, generator=g).cuda()
    (d2 * wgt).sum().backward()
    xr = pe.clone().requires_grad_(True)
    (O.lpg_forward(xr, 8)[0] * wgt.cpu()).sum().backward()
    assert (x.grad.cpu() - xr.grad).abs().max().item() <= 2e-5 * xr.grad.abs().max().item()
    # on a side stream: the operator launches on the CURRENT stream (the result is ready after that stream syncs)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        big = (torch.rand((8, 4, 44, 152), device="cuda") + 0.5)
        ds, _ = tops.lpg(big, 8)
    st.synchronize()
    assert torch.equal(ds, tops.lpg(big, 8)[0])
    # --- validation
    for bad, msg in ((lambda: tops.lpg(pe.cuda().double(), 4), "float32"), (lambda: tops.lpg(pe.cuda()[:, :3], 4), "[B,4,h,w]"),
                     (lambda: tops.lpg(pe.cuda(), 3), "upratio"),
                     (lambda: tops.reduction_1x1(torch.zeros((10, 32), device="cuda"), 32, 16, torch.zeros(8, device="cuda"), 80.0, True, True,
                                                 torch.zeros(9, device="cuda")), "out must hold"),
                     (lambda: tops.reduction_1x1(torch.zeros((10, 32), device="cuda"), 32, 16, torch.zeros(8, device="cuda"), 80.0, True, True,
                                                 torch.zeros(10)), "CUDA")):
        with pytest.raises(RuntimeError) as ei:
            bad()
        assert msg in str(ei.value), (msg, str(ei.value)[:200])
    with pytest.raises(RuntimeError) as ei:                                   # a chain that is not built: the C ABI's code, as text
        tops.reduction_1x1(torch.zeros((64, 48), device="cuda"), 48, 24, torch.zeros(64, device="cuda"), 80.0, False, True,
                           torch.zeros(256, device="cuda"))
    assert "bts_hip::reduction_1x1 failed" in str(ei.value)
    # --- the whole decoder through both bindings: bit-identical (the ctypes run in a child process: the binding is chosen at import)
    from parity_util import build_hip_decoder, hip_run
    got = hip_run(build_hip_decoder("K"), "K", 2, 64, 96, 4321)
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); from bts_amd import ops; assert ops.torch_ops() is None; "
            "from parity_util import build_hip_decoder, hip_run; torch.save([o.cpu() for o in hip_run(build_hip_decoder('K'), 'K', 2, 64, 96, 4321)], sys.argv[1])"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.pt")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, BTS_BINDING="ctypes"))
        other = torch.load(out, weights_only=True)
    assert all(torch.equal(a.cpu(), b) for a, b in zip(got, other)), "torch-op and ctypes bindings differ"


# ------------------------------------------------------------------------------- fused Winograd F(2x2,3x3) kernel (conv_wino.inc)
def _wino_case(cin, cout, shape, pre):
    import torch.nn.functional as F
    from bts_amd import ops
    B, h, w = shape
    g = torch.Generator().manual_seed(cin * 7 + cout + h)
    x = torch.randn((B, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(cin * 9.0)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    xin = torch.relu(x.double() * ps.double().view(1, -1, 1, 1) + pb.double().view(1, -1, 1, 1)) if pre else x.double()
    ref = F.elu(F.conv2d(xin, wt.double(), padding=1))
    x2d = x.permute(0, 2, 3, 1).reshape(B * h * w, cin).contiguous().cuda()
    wp = ops.pack_conv_weight(wt.cuda())[0]
    ybuf = torch.zeros((B * h * w, cout + 32), device="cuda")
    y = ybuf[:, 16:16 + cout]
    tr = ops.KernelTrace()
    ops.set_trace(tr)
    try:
        with ops.launch_config(fill_frames=16):
            ops.conv_forward(x2d, B, h, w, wp, cout, 3, act=ops.ACT_ELU, y2d=y, pre=(ps.cuda(), pb.cuda()) if pre else None, pre_relu=pre)
    finally:
        ops.set_trace(None)
    got = y.reshape(B, h, w, cout).permute(0, 3, 1, 2).cpu().double()
    assert float(ybuf[:, :16].abs().max()) == 0.0 and float(ybuf[:, 16 + cout:].abs().max()) == 0.0
    return sorted(tr.summary())[0], (got - ref).abs().max().item() / ref.abs().max().item(), y.cpu()


@fp32_only
@pytest.mark.parametrize("cin,cout,shape,pre", [(64, 128, (2, 44, 152), True), (448, 256, (1, 44, 152), False), (192, 48, (2, 88, 304), True),
                                                (128, 64, (1, 45, 150), True), (32, 128, (3, 15, 47), True)])
def test_winograd_kernel_vs_fp64_and_direct(cin, cout, shape, pre):
    """conv_wino_kernel (BTS_CONV_WINO=1): fused Winograd F(2x2,3x3) for stride-1 3x3 convolutions against torch in fp64,
    next to the direct kernel's own error on the same case; odd map sizes (ragged 8x16 workgroup tiles, odd last row /
    column of a 2x2 block), prologue + zero padding, ELU epilogue into a strided channel slice."""
    import os, subprocess, sys, tempfile
    if os.environ.get("BTS_CONV_WINO", "1") in ("", "0"):
        pytest.skip("direct kernels forced by BTS_CONV_WINO=0")
    kern, err_wino, y = _wino_case(cin, cout, shape, pre)
    assert kern == "conv_wino_kernel<%d>" % (128 if cout >= 128 else (48 if cout == 48 else 64)), kern      # 48: the 16x16x4-MFMA tile
    # the direct kernel on the same case, in a child process (the knob is read once per process)
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_round3_gpu as R; "
            "k, e, y = R._wino_case(%d, %d, %r, %r); assert not k.startswith('conv_wino'), k; torch.save((e, y), sys.argv[1])"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), cin, cout, shape, pre))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "y.pt")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, BTS_CONV_WINO="0"))
        err_direct, y_direct = torch.load(out, weights_only=True)
    print((cin, cout, shape), "max error / max|ref|: winograd %.3g, direct %.3g" % (err_wino, err_direct))
    assert err_wino <= max(1e-5, 4 * err_direct), (err_wino, err_direct)
    assert (y - y_direct).abs().max().item() / y_direct.abs().max().item() <= 2e-5
    B, h, w = shape
    if B > 1:                                   # frames are independent: the last frame alone gives the same bits
        from bts_amd import ops
        g = torch.Generator().manual_seed(cin * 7 + cout + h)
        x = torch.randn((B, cin, h, w), generator=g)
        wt = torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(cin * 9.0)
        ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
        x1 = x[B - 1:].permute(0, 2, 3, 1).reshape(h * w, cin).contiguous().cuda()
        wp = ops.pack_conv_weight(wt.cuda())[0]
        y1 = torch.zeros((h * w, cout), device="cuda")
        with ops.launch_config(fill_frames=16):
            ops.conv_forward(x1, 1, h, w, wp, cout, 3, act=ops.ACT_ELU, y2d=y1, pre=(ps.cuda(), pb.cuda()) if pre else None, pre_relu=pre)
        assert torch.equal(y1.cpu(), y[(B - 1) * h * w:])


@pytest.mark.parametrize("cout,cin_ld,n_tail,c16", [(64, 64, 0, 0), (96, 192, 0, 48), (128, 228, 1, 0), (64, 36, 4, 0)])
def test_pack_wino_weight_kernel_matches_the_torch_statement(cout, cin_ld, n_tail, c16):
    """bts_pack_wino_f32 (one launch) against pack_wino_weight_reference (torch, fp64 einsum): both fragment layouts, the
    planar-tail form (last 4 channels left out).  Both compute U = G g G^T in fp64 and round once; the summation order
    inside the double sums differs, so a value may land on the other side of a rounding tie: <= 1 ulp, almost all equal."""
    from bts_amd import ops
    g = torch.Generator().manual_seed(cout + cin_ld)
    kp = ops.round_up(9 * cin_ld, 32)
    wp = torch.zeros((cout, kp))
    wp[:, :9 * cin_ld] = torch.randn((cout, 9 * cin_ld), generator=g) * 0.1
    wp = wp.cuda()
    ref = ops.pack_wino_weight_reference(wp, cin_ld, n_tail, c_out16=c16)
    got = ops.pack_wino_weight(wp, cin_ld, n_tail, c_out16=c16)
    assert got.shape == ref.shape
    assert (got == ref).float().mean().item() > 0.9999
    assert (got - ref).abs().max().item() <= 1.2e-7 * ref.abs().max().item()
    again = torch.full_like(got, float("nan"))
    ops.pack_wino_weight(wp, cin_ld, n_tail, c_out16=c16, out=again)                   # refill in place
    assert torch.equal(again, got)
    with pytest.raises(Exception):
        ops.pack_wino_weight(wp, cin_ld, n_tail, c_out16=c16, out=again[:-4])


@fp32_only
def test_training_step_retransforms_the_winograd_weights():
    """The training step's packed weights are refilled IN PLACE every iteration (train.WeightPacker); forms derived from them
    and cached on the packed tensor (the Winograd U of ops.conv_forward) must follow.  Two forwards / input gradients of
    a Winograd-eligible 3x3 convolution with an optimiser-style in-place weight update in between: both match torch on the
    weights of THEIR step (a stale U would reproduce the first step's result)."""
    from bts_amd import ops, train
    torch.manual_seed(5)
    B, cin, cout, h, w = 4, 64, 64, 64, 128
    weight = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) * 0.05).cuda())
    x = torch.randn(B, cin, h, w).cuda().to(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(B, cout, h, w).cuda()

    def step():
        train.begin_step()
        tr = ops.KernelTrace()
        ops.set_trace(tr)
        try:
            y = train.conv2d(x, weight, padding=1, tag="t")
            (dx,) = torch.autograd.grad(y, x, gy)
        finally:
            ops.set_trace(None)
        assert any(n.startswith("conv_wino_kernel") for n in tr.summary()), list(tr.summary())
        yr = torch.nn.functional.conv2d(x.detach().double(), weight.detach().double(), padding=1)
        dxr = torch.nn.functional.conv_transpose2d(gy.double(), weight.detach().double(), padding=1)
        return ((y.double() - yr).abs().max() / yr.abs().max()).item(), ((dx.double() - dxr).abs().max() / dxr.abs().max()).item(), y.detach().clone()

    e1, d1, y1 = step()
    with torch.no_grad():                                   # what a fused optimiser does: new values, same storage
        weight.mul_(-0.5).add_(0.01)
    e2, d2, y2 = step()
    assert max(e1, d1, e2, d2) <= 5e-6, (e1, d1, e2, d2)
    assert (y2 - y1).abs().max().item() > 1e-2              # the two steps really differ


@pytest.mark.parametrize("cname,B,H,W", [("K", 2, 64, 96), ("N", 1, 96, 128)])
def test_bts_size_256_decoder_matches_the_oracle(cname, B, H, W):
    """params.bts_size = 256 (bts.py:198-217 derive every decoder width from it): reduction chains 64->64.. / 64->32.. /
    32->16.. / 16->8->1 on the narrow (16x16x4 MFMA) kernel, get_depth on 16 channels, every convolution at half width.
    Whole decoder vs the CPU oracle on the same PCG64 synthetic state, tolerances of the bts_size 512 tests."""
    from bts_amd import bts as M
    from parity_util import CONFIGS, check_outputs, make_inputs
    from oracle import bts_oracle as O
    enc, md, ds, _, _ = CONFIGS[cname]
    feat = synth.ENCODER_CHANNELS[enc]
    state_np = synth.decoder_state(feat, 256, 0)
    dec = M.bts(Params(enc, 256, md, ds), feat, 256)
    dec.load_state_dict({k: (torch.tensor(v) if np.ndim(v) == 0 else t(v)) for k, v in state_np.items()}, strict=True)
    dec = dec.eval().cuda()
    feats, focal = make_inputs(cname, B, H, W, 11)
    with torch.no_grad():
        ref, inter = O.decoder_forward(O.state_from_numpy(state_np), feats, focal, md, ds, want_intermediates=True)
        got = dec([None] + [f.cuda() for f in feats[1:]], focal.cuda())
    torch.cuda.synchronize()
    rep = check_outputs(got, ref, inter, what="bts_size 256 %s" % cname)
    assert max(v for k, v in rep.items() if k != "iconv1_max_abs") <= 2e-5, rep
