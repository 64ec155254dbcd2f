"""Training step on the HIP kernels: convolution input/weight gradients vs torch
autograd on the CPU, the decoder's train()-mode step vs the reference-generated golden (decoder_train.npz) and the
CPU oracle, and one whole-model step (DenseNet encoder + decoder)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bts_amd import synth
from oracle import bts_oracle as O
from parity_util import (CONFIGS, TRAIN_CASE, Params, assert_grads_close, check_train_against_golden, grad_error_report,
                         make_inputs, oracle_train_step, t)

from conftest import fp32_only

pytestmark = pytest.mark.gpu


def _ref_conv(x, w, stride, padding, dilation, up):
    if up == 2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, w, stride=stride, padding=padding, dilation=dilation)


CONV_CASES = [
    # B, cin, cout, h, w, k, dil, pad, stride, up, x_needs_grad
    (2, 128, 128, 9, 13, 1, 1, 0, 1, 1, True),        # reduc / aspp 1x1
    (2, 256, 128, 11, 19, 3, 3, 3, 1, 1, True),       # daspp_3
    (1, 64, 32, 13, 17, 3, 24, 24, 1, 1, True),       # dilation larger than the map
    (2, 225, 128, 10, 14, 3, 1, 1, 1, 1, True),       # conv3: odd channel count
    (2, 8, 3, 16, 24, 1, 1, 0, 1, 1, True),           # plane_params
    (2, 8, 1, 16, 24, 1, 1, 0, 1, 1, True),           # final
    (2, 36, 32, 12, 20, 3, 1, 1, 1, 1, True),         # conv1
    (2, 48, 40, 6, 9, 3, 1, 1, 1, 2, True),           # upconv: nearest-2x folded in
    (2, 3, 96, 32, 48, 7, 1, 3, 2, 1, False),         # densenet conv0 (stride 2; the image needs no gradient)
    (3, 192, 48, 7, 11, 3, 1, 1, 1, 1, True),         # densenet growth conv
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d_gradients_vs_torch_cpu(case):
    from bts_amd import train
    B, cin, cout, h, w, k, dil, pad, stride, up, xg = case
    gen = torch.Generator().manual_seed(hash(case) % (1 << 31))
    x = torch.randn(B, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, k, k, generator=gen) / np.sqrt(cin * k * k)
    x64, w64 = x.double().requires_grad_(xg), wt.double().requires_grad_(True)
    y_ref = _ref_conv(x64, w64, stride, pad, dil, up)
    gy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(gy.double())
    xd, wd = x.cuda().requires_grad_(xg), wt.cuda().requires_grad_(True)
    y = train.conv2d(xd, wd, padding=pad, dilation=dil, stride=stride, up=up)
    assert tuple(y.shape) == tuple(y_ref.shape)
    y.backward(gy.cuda())
    torch.cuda.synchronize()

    def close(got, ref, what):
        ref = ref.float()
        scale = ref.abs().max().item()
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 2e-5 * scale * np.sqrt(max(1, ref.numel() // ref.shape[0] // 64)) + 1e-6, (what, err, scale)

    close(y.detach(), y_ref.detach(), "forward")
    close(wd.grad, w64.grad, "weight grad")
    if xg:
        close(xd.grad, x64.grad, "input grad")


GROUPED_CASES = [
    # B, cin, cout, h, w, k, pad, stride, groups
    (2, 128, 128, 10, 14, 3, 1, 1, 32),        # resnext50 layer1: 4 channels per group (8 groups per 32-channel bundle)
    (2, 256, 256, 10, 14, 3, 1, 2, 32),        # 8 per group, strided (first block of a stage)
    (2, 512, 512, 6, 8, 3, 1, 1, 32),          # 16 per group
    (1, 1024, 1024, 6, 8, 3, 1, 2, 32),        # 32 per group = one group per bundle, strided
    (1, 2048, 2048, 3, 4, 3, 1, 1, 32),        # 64 per group (64-channel bundles)
    (2, 128, 128, 10, 14, 3, 1, 2, 1),         # resnet50 conv2, dense, strided
    (2, 256, 512, 10, 14, 1, 0, 2, 1),         # downsample 1x1 stride 2
]


@pytest.mark.parametrize("case", GROUPED_CASES, ids=lambda c: "x".join(map(str, c)))
def test_grouped_and_strided_conv_gradients_vs_torch_cpu(case):
    """ResNet / ResNeXt bottleneck convolutions in the training graph: channel-bundled grouped 3x3 (forward, input and
    weight gradient) and the stride-2 input gradient (zero-inserted adjoint) vs torch autograd in fp64."""
    from bts_amd import train
    B, cin, cout, h, w, k, pad, stride, groups = case
    gen = torch.Generator().manual_seed(cin + 13 * stride + groups)
    x = torch.randn(B, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin // groups, k, k, generator=gen) / np.sqrt(cin // groups * k * k)
    x64, w64 = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y_ref = F.conv2d(x64, w64, stride=stride, padding=pad, groups=groups)
    gy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(gy.double())
    xd, wd = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True)
    y = train.conv2d(xd, wd, padding=pad, stride=stride, groups=groups)
    assert tuple(y.shape) == tuple(y_ref.shape)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    for got, ref, what in ((y.detach(), y_ref.detach(), "forward"), (wd.grad, w64.grad, "weight grad"), (xd.grad, x64.grad, "input grad")):
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 5e-5 * scale, (what, err, scale)


def test_wgrad_input_prologue_matches_materialised_input():
    """bts_conv_wgrad_desc.pre_*: gathering relu(x*scale + shift) on the fly == the weight gradient on the materialised
    tensor (zero padding stays zero: a shift must not leak into the border taps)."""
    from bts_amd import ops
    B, h, w, cin, cout = 2, 9, 13, 64, 32
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(B * h * w, cin, generator=gen).cuda()
    dy = torch.randn(B * h * w, cout, generator=gen).cuda()
    sc = (torch.rand(cin, generator=gen) + 0.5).cuda()
    sh = (torch.randn(cin, generator=gen) * 0.5 + 0.3).cuda()
    ws = torch.empty(4 << 20, device="cuda")
    mat = torch.relu(x * sc + sh)
    ref = ops.conv_wgrad(mat, B, h, w, cin, dy, cout, 3, ws=ws)
    got = ops.conv_wgrad(x, B, h, w, cin, dy, cout, 3, ws=ws, pre=(sc, sh), pre_relu=True)
    assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    ref1 = ops.conv_wgrad(x * sc + sh, B, h, w, cin, dy, cout, 1, ws=ws)
    got1 = ops.conv_wgrad(x, B, h, w, cin, dy, cout, 1, ws=ws, pre=(sc, sh), pre_relu=False)
    assert (got1 - ref1).abs().max().item() <= 1e-5 * ref1.abs().max().item()


def test_wgrad_split_is_deterministic_and_matches_unsplit():
    """The pixel split only regroups a sum: with and without workspace agree to fp32 rounding, and two runs of the
    split path are bit-identical (fixed-order reduction, no atomics)."""
    from bts_amd import ops
    B, h, w, cin, cout = 4, 44, 76, 64, 32
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B * h * w, cin, generator=gen).cuda()
    dy = torch.randn(B * h * w, cout, generator=gen).cuda()
    ws = torch.empty(8 << 20, device="cuda")
    a = ops.conv_wgrad(x, B, h, w, cin, dy, cout, 3, ws=ws)
    b = ops.conv_wgrad(x, B, h, w, cin, dy, cout, 3, ws=ws)
    c = ops.conv_wgrad(x, B, h, w, cin, dy, cout, 3, ws=None)
    assert torch.equal(a, b)
    assert (a - c).abs().max().item() <= 1e-4 * c.abs().max().item()


@pytest.mark.parametrize("shape,relu", [((2, 48, 5, 7), True), ((3, 128, 9, 13), False), ((2, 2208, 2, 3), True),
                                        ((4, 96, 64, 96), True), ((1, 256, 1, 2), False), ((2, 132, 16, 24), True),
                                        ((2, 32, 33, 41), False), ((1, 64, 176, 352), True)])
def test_bn_train_forward_backward_vs_torch(shape, relu):
    """Batch-statistic BN (+fused ReLU) kernels vs F.batch_norm(training=True) in fp64 on the CPU: output, running
    statistics (momentum 0.01, unbiased variance), and all three gradients; also on a channel slice of a wider
    buffer (row stride > C), the way concat buffers hand their slices over."""
    from bts_amd import train
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(C * 131 + H)
    wide = torch.randn(B, C + 8, H, W, generator=gen) * 1.7 + 0.4
    x = wide[:, 4:4 + C]
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.1
    rm, rv = torch.randn(C, generator=gen) * 0.1, torch.rand(C, generator=gen) + 0.5
    x64, g64, b64 = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y64 = F.batch_norm(x64, rm64, rv64, g64, b64, True, 0.01, 1.1e-5)
    if relu:
        y64 = F.relu(y64)
    gy = torch.randn(shape, generator=gen)
    y64.backward(gy.double())
    bn = torch.nn.BatchNorm2d(C, eps=1.1e-5, momentum=0.01)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn = bn.cuda().train()
    wide_d = wide.cuda().contiguous(memory_format=torch.channels_last)
    xd = wide_d[:, 4:4 + C].detach().requires_grad_(True)
    y = train._bn(xd, bn, relu=relu)
    y.backward(gy.cuda())
    torch.cuda.synchronize()

    def close(got, ref, tol, what):
        scale = max(ref.abs().max().item(), 1e-12)
        err = (got.detach().cpu().double() - ref).abs().max().item()
        assert err <= tol * scale, (what, err, scale)
    close(y, y64.detach(), 2e-5, "forward")
    close(bn.running_mean, rm64, 1e-5, "running_mean")
    close(bn.running_var, rv64, 1e-5, "running_var")
    assert int(bn.num_batches_tracked.item()) == 1
    close(xd.grad, x64.grad, 2e-4, "dx")
    close(bn.weight.grad, g64.grad, 2e-4, "dgamma")
    close(bn.bias.grad, b64.grad, 2e-4, "dbeta")


def _train_decoder(device="cuda"):
    from bts_amd import bts as M
    c = TRAIN_CASE
    enc, md, ds, _, _ = CONFIGS[c["cname"]]
    feat = synth.ENCODER_CHANNELS[enc]
    dec = M.bts(Params(enc, 512, md, ds), feat, 512)
    sd = {k: (torch.tensor(v) if np.ndim(v) == 0 else t(v)) for k, v in synth.decoder_state(feat, 512, 0).items()}
    dec.load_state_dict(sd, strict=True)
    return dec.train().to(device)


def test_decoder_train_step_vs_golden_and_oracle(golden_dir):
    """bts.forward in train() mode + silog loss + backward on the GPU == the reference's own training step."""
    from bts_amd import bts as M
    c = TRAIN_CASE
    _, md, ds, _, _ = CONFIGS[c["cname"]]
    g = np.load(os.path.join(golden_dir, "decoder_train.npz"))
    dec = _train_decoder()
    feats, focal = make_inputs(c["cname"], c["B"], c["H"], c["W"], c["feat_seed"])
    feats = [None] + [f.cuda().requires_grad_(True) for f in feats[1:]]
    gt, mask = synth.train_targets(c["B"], c["H"], c["W"], md, c["target_seed"])
    outs = dec(feats, focal.cuda())
    loss = M.silog_loss(variance_focus=c["variance_focus"])(outs[4], t(gt).cuda(), t(mask).cuda())
    loss.backward()
    torch.cuda.synchronize()
    pg = {n: p.grad.cpu().numpy() for n, p in dec.named_parameters()}
    bufs = {n: b.cpu().numpy() for n, b in dec.named_buffers() if n.endswith(("running_mean", "running_var"))}
    check_train_against_golden(g, loss.item(), [o.detach().cpu().numpy() for o in outs],
                               [f.grad.cpu().numpy() for f in feats[1:]], pg, bufs, what="hip/golden", robust=True)
    np.testing.assert_allclose([dec.lpg8x8.abs_min.item(), dec.lpg4x4.abs_min.item(), dec.lpg2x2.abs_min.item()],
                               g["abs_min"], rtol=1e-3, atol=1e-6)
    # every element of every gradient against the oracle in fp64 (the exact-arithmetic yardstick), with the fp32 CPU
    # oracle's own distance to it printed beside ours
    r64, r32 = oracle_train_step(torch.float64), oracle_train_step()
    ref = {n: v.numpy() for n, v in r64["param_grads"].items()}
    per, l2 = grad_error_report(pg, ref)
    per32, l2_32 = grad_error_report({n: v.numpy() for n, v in r32["param_grads"].items()}, ref)
    print("global rel-L2 vs fp64: hip %.2e, cpu fp32 oracle %.2e; worst tensor hip %.2e, cpu fp32 %.2e"
          % (l2, l2_32, max(per.values()), max(per32.values())))
    assert_grads_close(per, l2, "hip/fp64 oracle", fp32_floor=(per32, l2_32))


def test_train_mode_modules_match_oracle():
    """Module-level train() forwards (atrous_conv with batch-stat BN, reduction_1x1, upconv) vs the oracle."""
    from bts_amd import bts as M
    torch.manual_seed(5)
    x = torch.randn(2, 64, 11, 19)
    m = M.atrous_conv(64, 32, 6).train()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.2 + (1.0 if p.dim() == 1 else 0.0))
    p = {"d.atrous_conv." + k: v.detach().clone() for k, v in m.atrous_conv.state_dict().items()}
    ref = O.atrous_forward(x, p, "d", 6, True, training=True)
    got = m.cuda()(x.cuda())
    assert (got.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    np.testing.assert_allclose(m.atrous_conv.first_bn.running_mean.cpu().numpy(),
                               p["d.atrous_conv.first_bn.running_mean"].numpy(), rtol=1e-4, atol=1e-6)
    r = M.reduction_1x1(64, 32, 80.0).train()
    ws = [mm.weight.detach().clone() for mm in r.reduc.modules() if isinstance(mm, torch.nn.Conv2d)]
    ref = O.reduction_forward(x, ws, 80.0, False)
    got = r.cuda()(x.cuda())
    assert (got.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    u = M.upconv(64, 16).train()
    ref = O.upconv_forward(x, u.conv.weight.detach())
    got = u.cuda()(x.cuda())
    assert (got.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@fp32_only
def test_btsmodel_train_step_densenet121_vs_cpu():
    """One whole-model training step (bts_main.py:476-500 protocol): DenseNet121 encoder + decoder, silog loss,
    backward.  CPU side: the same torch encoder modules + the oracle decoder with autograd, in fp64 (yardstick) and
    fp32 (the noise floor fp32 arithmetic itself has on this step, see parity_util.fp32_noise_floor)."""
    import copy
    from bts_amd import bts as M
    params = Params("densenet121_bts", 512, 80.0, "kitti")
    torch.manual_seed(21)
    model = M.BtsModel(params).train()
    B, H, W = 2, 64, 96
    x = torch.from_numpy(synth.image_batch(B, H, W, 5))
    focal = torch.from_numpy(synth.focal_values(B, "kitti", 5))
    gt, mask = synth.train_targets(B, H, W, 80.0, 9)

    def cpu_step(dtype):
        enc = copy.deepcopy(model.encoder).to(dtype)
        state = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone())
                 for k, v in model.decoder.state_dict().items()}
        for k, v in state.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        outs = O.decoder_forward(state, enc(x.to(dtype)), focal.to(dtype), 80.0, "kitti", training=True)
        loss = O.silog_loss(outs[4], t(gt).to(dtype), t(mask), 0.85)
        loss.backward()
        grads = {"encoder." + n: p.grad for n, p in enc.named_parameters()}
        grads.update({"decoder." + n: v.grad for n, v in state.items() if v.requires_grad})
        return loss.item(), grads

    loss64, g64 = cpu_step(torch.float64)
    loss32, g32 = cpu_step(torch.float32)
    mg = model.cuda()
    outs = mg(x.cuda(), focal.cuda())
    loss = M.silog_loss(0.85)(outs[4], t(gt).cuda(), t(mask).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss64) <= max(2e-4 * abs(loss64), 4 * abs(loss32 - loss64)), (loss.item(), loss64, loss32)
    got = {n: p.grad.cpu().numpy() for n, p in mg.named_parameters()}
    ref = {n: v.numpy() for n, v in g64.items()}
    per, l2 = grad_error_report(got, ref)
    per32, l2_32 = grad_error_report({n: v.numpy() for n, v in g32.items()}, ref)
    print("whole-model step, global rel-L2 vs fp64: hip %.2e, cpu fp32 %.2e; worst tensor hip %.2e, cpu fp32 %.2e"
          % (l2, l2_32, max(per.values()), max(per32.values())))
    assert_grads_close(per, l2, "whole model / fp64", fp32_floor=(per32, l2_32))
    checked = len(per)
    assert checked > 400


def test_trainer_steps_update_only_trainable_parameters():
    """Three iterations of the bts_main.py protocol (set_misc freeze, AdamW groups, poly LR, silog on valid pixels):
    frozen encoder layers stay bit-identical, everything else moves, running statistics advance, loss stays finite."""
    from bts_amd import bts as M, trainer
    params = Params("densenet121_bts", 512, 80.0, "kitti")
    torch.manual_seed(3)
    model = M.BtsModel(params).train().cuda()
    frozen = set(trainer.set_misc(model, params.encoder))
    opt = trainer.make_optimizer(model, 1e-4, 1e-2, 1e-3)
    crit = M.silog_loss(0.85)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    B, H, W = 2, 64, 96
    x = torch.from_numpy(synth.image_batch(B, H, W, 5)).cuda()
    focal = torch.from_numpy(synth.focal_values(B, "kitti", 5)).cuda()
    gt, _ = synth.train_targets(B, H, W, 80.0, 9)
    gt = torch.from_numpy(gt).cuda()
    losses = []
    for step in range(3):
        loss, outs = trainer.train_step(model, opt, crit, x, focal, gt, lr=trainer.poly_lr(step, 100, 1e-4))
        losses.append(loss.item())
    assert all(np.isfinite(losses)), losses
    assert len(outs) == 6 and tuple(outs[4].shape) == (B, 1, H, W)
    for n, p in model.named_parameters():
        same = torch.equal(p.detach(), before[n])
        if n[len("encoder."):] in frozen:
            assert same and p.grad is None, n
        else:
            assert not same, n
    assert int(model.decoder.bn5.num_batches_tracked.item()) == 3
    assert model.decoder.lpg8x8.abs_min is not None


ADJOINT_CASES = [
    # the reference training crop (352x704) and the KITTI test size (352x1216): name, B, cin, cout, h, w, k, dil, up
    ("conv1 @352x1216", 1, 36, 32, 352, 1216, 3, 1, 1),
    ("upconv1 @176x608 -> 352x1216", 1, 64, 32, 176, 608, 3, 1, 2),
    ("conv2 @176x352", 4, 164, 64, 176, 352, 3, 1, 1),
    ("daspp_24 3x3 @44x152", 2, 256, 128, 44, 152, 3, 24, 1),
    ("daspp_6 1x1 @44x88", 4, 576, 256, 44, 88, 1, 1, 1),
    ("upconv5 @11x22 -> 22x44", 4, 2208, 512, 11, 22, 3, 1, 2),
    ("reduc1x1 16->8 @352x704", 4, 16, 8, 352, 704, 1, 1, 1),
]


@pytest.mark.parametrize("case", ADJOINT_CASES, ids=lambda c: c[0])
def test_conv_gradients_are_adjoints_at_full_size(case):
    """Size-independent property at BASELINE sizes (no CPU reference needed): a convolution is bilinear, so for any
    x, w, dy:  <conv(x, w), dy>  ==  <x, dgrad(dy)>  ==  <w, wgrad(x, dy)>.  All three inner products come from
    three different kernels/launch shapes; they must agree to fp32 accumulation accuracy."""
    from bts_amd import train
    _, B, cin, cout, h, w, k, dil, up = case
    gen = torch.Generator(device="cuda").manual_seed(cin * 7 + cout)
    x = torch.randn(B, cin, h, w, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(cout, cin, k, k, device="cuda", generator=gen) / np.sqrt(cin * k * k)).requires_grad_(True)
    y = train.conv2d(x, wt, padding=dil * (k // 2), dilation=dil, up=up)
    dy = torch.randn(y.shape, device="cuda", generator=gen)
    y.backward(dy)
    torch.cuda.synchronize()
    a = (y.detach().double() * dy.double()).sum().item()
    b = (x.detach().double() * x.grad.double()).sum().item()
    c = (wt.detach().double() * wt.grad.double()).sum().item()
    scale = np.sqrt(float(y.numel()))                       # |<y,dy>| ~ sqrt(n) for independent unit-variance entries
    assert abs(a - b) <= 2e-4 * scale + 1e-5 * abs(a), (a, b, scale)
    assert abs(a - c) <= 2e-4 * scale + 1e-5 * abs(a), (a, c, scale)


def test_decoder_train_step_full_training_crop_vs_oracle():
    """Decoder training step at the reference's training crop (352x704, arguments_train_eigen.txt) for one frame pair
    against the CPU oracle in fp32: loss, outputs and the global gradient in relative L2."""
    from bts_amd import bts as M
    B, H, W = 2, 352, 704
    enc, md, ds, _, _ = CONFIGS["K"]
    feat = synth.ENCODER_CHANNELS[enc]
    state = O.state_from_numpy(synth.decoder_state(feat, 512, 0))
    for k, v in state.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    feats, focal = make_inputs("K", B, H, W, 77)
    gt, mask = synth.train_targets(B, H, W, md, 78)
    outs_ref = O.decoder_forward(state, feats, focal, md, ds, training=True)
    loss_ref = O.silog_loss(outs_ref[4], t(gt), t(mask), 0.85)
    loss_ref.backward()
    dec = _train_decoder()
    outs = dec([None] + [f.cuda() for f in feats[1:]], focal.cuda())
    loss = M.silog_loss(0.85)(outs[4], t(gt).cuda(), t(mask).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()), (loss.item(), loss_ref.item())
    fd, fr = outs[4].detach().cpu(), outs_ref[4].detach()
    assert ((fd - fr).abs() / fr.abs().clamp_min(1e-3)).max().item() <= 1e-3
    got = {n: p.grad.cpu().numpy() for n, p in dec.named_parameters()}
    ref = {n: v.grad.numpy() for n, v in state.items() if v.requires_grad}
    per, l2 = grad_error_report(got, ref)
    print("352x704 decoder step vs CPU fp32 oracle: global rel-L2 %.2e, worst tensor %.2e, 90th pct %.2e"
          % (l2, max(per.values()), sorted(per.values())[int(0.9 * (len(per) - 1))]))
    # fp32 against fp32 (both sides carry their own rounding / ReLU-mask flips; per-tensor maxima over up to 10 M
    # elements): per-tensor bars are loose, the global relative L2 -- measured 3.3e-4 -- is the tight one
    # both sides are fp32 here (two independent noise sources), so large tensors get 3e-2 where a comparison against
    # fp64 gets 2e-2 (measured worst large tensor: 2.0e-2, daspp_6's 1x1 weight)
    assert_grads_close(per, l2, "352x704 decoder step", typical=1e-2, worst=0.1, l2=2e-3, worst_large=3e-2)


def test_btsmodel_train_step_resnext50_vs_cpu():
    """Whole-model training step with a ResNeXt encoder (grouped + strided convolutions in the graph) vs the CPU:
    torch encoder modules + oracle decoder, fp64 yardstick and the fp32 CPU run as the noise floor."""
    import copy
    from bts_amd import bts as M
    params = Params("resnext50_bts", 512, 10.0, "nyu")
    torch.manual_seed(31)
    model = M.BtsModel(params).train()
    B, H, W = 2, 64, 96
    x = torch.from_numpy(synth.image_batch(B, H, W, 15))
    focal = torch.from_numpy(synth.focal_values(B, "nyu", 15))
    gt, mask = synth.train_targets(B, H, W, 10.0, 19)

    def cpu_step(dtype):
        enc = copy.deepcopy(model.encoder).to(dtype)
        state = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone())
                 for k, v in model.decoder.state_dict().items()}
        for k, v in state.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        outs = O.decoder_forward(state, enc(x.to(dtype)), focal.to(dtype), 10.0, "nyu", training=True)
        loss = O.silog_loss(outs[4], t(gt).to(dtype), t(mask), 0.85)
        loss.backward()
        grads = {"encoder." + n: p.grad for n, p in enc.named_parameters() if p.grad is not None}
        grads.update({"decoder." + n: v.grad for n, v in state.items() if v.requires_grad})
        return loss.item(), grads

    loss64, g64 = cpu_step(torch.float64)
    loss32, g32 = cpu_step(torch.float32)
    mg = model.cuda()
    outs = mg(x.cuda(), focal.cuda())
    loss = M.silog_loss(0.85)(outs[4], t(gt).cuda(), t(mask).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss64) <= max(2e-4 * abs(loss64), 4 * abs(loss32 - loss64)), (loss.item(), loss64, loss32)
    got = {n: p.grad.cpu().numpy() for n, p in mg.named_parameters() if p.grad is not None}
    assert set(got) == set(g64), set(got) ^ set(g64)          # the unused fc head gets no gradient on either side
    ref = {n: v.numpy() for n, v in g64.items()}
    per, l2 = grad_error_report(got, ref)
    per32, l2_32 = grad_error_report({n: v.numpy() for n, v in g32.items()}, ref)
    print("ResNeXt50 whole-model step, global rel-L2 vs fp64: hip %.2e, cpu fp32 %.2e; worst tensor hip %.2e, cpu fp32 %.2e"
          % (l2, l2_32, max(per.values()), max(per32.values())))
    assert_grads_close(per, l2, "resnext50 whole model / fp64", fp32_floor=(per32, l2_32))


def test_fused_dense_block_equals_layer_by_layer_graph():
    """train._DenseBlockFn (one in-place autograd node per DenseNet block, norm+ReLU in the conv / wgrad prologues) and
    the generic layer-by-layer graph (torch.cat, separate BN nodes) are the same function: loss, running statistics
    and every gradient agree to fp32 re-association level."""
    import copy
    from bts_amd import bts as M, train
    params = Params("densenet121_bts", 512, 80.0, "kitti")
    torch.manual_seed(77)
    base = M.BtsModel(params).train()
    B, H, W = 2, 64, 96
    x = torch.from_numpy(synth.image_batch(B, H, W, 3)).cuda()
    focal = torch.from_numpy(synth.focal_values(B, "kitti", 3)).cuda()
    gt, mask = synth.train_targets(B, H, W, 80.0, 4)
    gt, mask = t(gt).cuda(), t(mask).cuda()
    res = {}
    for fused in (True, False):
        m = copy.deepcopy(base).cuda()
        prev = train.FUSED_DENSE_BLOCKS
        train.FUSED_DENSE_BLOCKS = fused
        try:
            outs = m(x, focal)
            loss = M.silog_loss(0.85)(outs[4], gt, mask)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            train.FUSED_DENSE_BLOCKS = prev
        res[fused] = (loss.item(), {n: p.grad.cpu().numpy() for n, p in m.named_parameters()},
                      {n: b.cpu().numpy() for n, b in m.named_buffers() if "running" in n})
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * abs(res[False][0])
    for n, v in res[False][2].items():
        np.testing.assert_allclose(res[True][2][n], v, rtol=1e-4, atol=1e-6, err_msg=n)
    per, l2 = grad_error_report(res[True][1], res[False][1])
    print("fused vs layer-by-layer: global rel-L2 %.2e, worst tensor %.2e" % (l2, max(per.values())))
    # same kernels, same inputs, only the association of a few sums differs: measured 4.7e-7 global, 5.5e-6 worst
    assert_grads_close(per, l2, "fused dense block vs generic graph", typical=1e-4, worst=2e-3, l2=1e-4)


def test_training_graph_with_frozen_batchnorm_layers():
    """bts_main.py --bn_no_track_stats applies bn_init_as_tf (norm layers in eval mode inside a train()-mode model):
    the graph then normalises with the running statistics, leaves the buffers untouched, and still back-propagates."""
    from bts_amd import bts as M, trainer
    params = Params("densenet121_bts", 512, 80.0, "kitti")
    torch.manual_seed(5)
    model = M.BtsModel(params).train().cuda()
    trainer.set_misc(model, params.encoder, bn_no_track_stats=True)
    assert not model.decoder.bn5.training and model.decoder.conv5.training
    before = {n: b.clone() for n, b in model.named_buffers() if "running" in n}
    B, H, W = 1, 64, 96
    x = torch.from_numpy(synth.image_batch(B, H, W, 2)).cuda()
    focal = torch.from_numpy(synth.focal_values(B, "kitti", 2)).cuda()
    gt, mask = synth.train_targets(B, H, W, 80.0, 2)
    outs = model(x, focal)
    loss = M.silog_loss(0.85)(outs[4], t(gt).cuda(), t(mask).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(loss.item())
    for n, b in model.named_buffers():
        if "running" in n:
            assert torch.equal(b, before[n]), n
    g = model.decoder.conv5[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max().item() > 0
    assert model.encoder.base_model.denseblock2.denselayer3.conv2.weight.grad is not None
