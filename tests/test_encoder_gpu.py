"""GPU: the DenseNet encoder on the HIP conv kernel (bts_amd.encoder_hip) against the same nn modules
run by PyTorch on the CPU, and the fused BtsModel.forward against CPU encoder + oracle decoder."""
import os

import numpy as np
import pytest
import torch

from bts_amd import synth
from oracle import bts_oracle as O
from parity_util import Params, check_outputs

pytestmark = pytest.mark.gpu


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.empty_like(m.weight).uniform_(0.8, 1.2, generator=g)
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.05
            m.running_mean = torch.randn(m.running_mean.shape, generator=g) * 0.05
            m.running_var = torch.empty_like(m.running_var).uniform_(0.8, 1.2, generator=g)


@pytest.mark.parametrize("enc,shape", [("densenet161_bts", (2, 64, 96)), ("densenet121_bts", (1, 96, 64))])
def test_densenet_hip_taps_vs_torch_cpu(enc, shape):
    from bts_amd import bts as M
    from bts_amd.encoder_hip import DenseNetHip
    torch.manual_seed(3)
    e = M.encoder(Params(enc, 512, 80.0, "kitti")).eval()
    _randomise_bn(e, 5)
    B, H, W = shape
    x = torch.from_numpy(synth.image_batch(B, H, W, 21))
    with torch.no_grad():
        ref = e(x)
    eg = M.encoder(Params(enc, 512, 80.0, "kitti")).eval()
    eg.load_state_dict(e.state_dict())
    eg = eg.cuda()
    plan = DenseNetHip(eg.base_model)
    with torch.no_grad():
        got = plan.taps_nchw(x.cuda())
    assert len(got) == len(ref) == 6
    for i in range(1, 6):
        r, g = ref[i], got[i].cpu()
        assert r.shape == g.shape
        scale = r.abs().max().item()
        err = (r - g).abs().max().item()
        assert err <= 2e-4 * scale + 1e-5, "tap %d: max abs err %g (scale %g)" % (i, err, scale)


def test_btsmodel_fused_forward_vs_cpu():
    """BtsModel.forward (HIP encoder + HIP decoder, taps written in place) == CPU torch encoder + oracle decoder."""
    from bts_amd import bts as M
    params = Params("densenet161_bts", 512, 80.0, "kitti")
    torch.manual_seed(11)
    model = M.BtsModel(params).eval()
    _randomise_bn(model.encoder, 7)
    feat = synth.ENCODER_CHANNELS[params.encoder]
    state_np = synth.decoder_state(feat, 512, 0)
    model.decoder.load_state_dict({k: (torch.tensor(v) if np.ndim(v) == 0 else torch.from_numpy(v.copy())) for k, v in state_np.items()})
    B, H, W = 2, 64, 96
    x = torch.from_numpy(synth.image_batch(B, H, W, 5))
    focal = torch.from_numpy(synth.focal_values(B, "kitti", 5))
    with torch.no_grad():
        feats = model.encoder(x)
        ref_outs, inter = O.decoder_forward(O.state_from_numpy(state_np), feats, focal, 80.0, "kitti", want_intermediates=True)
    mg = M.BtsModel(params).eval()
    mg.load_state_dict(model.state_dict())
    mg = mg.cuda()
    with torch.no_grad():
        got = mg(x.cuda(), focal.cuda())
        mg.native_encoder = False                      # torch encoder (NCHW taps) + HIP decoder: same answer
        got2 = mg(x.cuda(), focal.cuda())
    rep = check_outputs(got, ref_outs, inter, rel_tol=5e-4, what="fused BtsModel")
    print("fused model max-rel:", rep)
    check_outputs(got2, ref_outs, inter, rel_tol=5e-4, what="torch-encoder BtsModel")


def test_sub_batch_streams_bit_identical():
    """BtsModel.sub_batches (concurrent half-batches on two streams, outputs written in place) must not change
    a single bit, nor the batch-wide abs_min telemetry."""
    from bts_amd import bts as M
    params = Params("densenet161_bts", 512, 80.0, "kitti")
    torch.manual_seed(2)
    m = M.BtsModel(params).eval().cuda()
    x = torch.from_numpy(synth.image_batch(4, 64, 96, 9)).cuda()
    focal = torch.from_numpy(synth.focal_values(4, "kitti", 9)).cuda()
    with torch.no_grad():
        m.sub_batches = 1
        a = [o.clone() for o in m(x, focal)]
        am_a = [m.decoder.lpg8x8.abs_min.item(), m.decoder.lpg4x4.abs_min.item(), m.decoder.lpg2x2.abs_min.item()]
        for S in (2, 4):
            m.sub_batches = S
            b = m(x, focal)
            torch.cuda.synchronize()
            for i in range(6):
                assert torch.equal(a[i], b[i]), "output %d differs with %d sub-batches" % (i, S)
            am_b = [m.decoder.lpg8x8.abs_min.item(), m.decoder.lpg4x4.abs_min.item(), m.decoder.lpg2x2.abs_min.item()]
            assert am_a == am_b
        m.sub_batches = 3                      # 4 % 3 != 0 -> largest divisor below (2)
        c = m(x, focal)
        assert all(torch.equal(a[i], c[i]) for i in range(6))


def test_bts_test_py_call_protocol(tmp_path):
    """The reference inference driver's sequence (bts_test.py:70-76, 90-102, 127-138) against the plugin module:
    import-by-name from the checkpoint directory, BtsModel(params=args), DataParallel, load_state_dict of a
    'module.'-prefixed checkpoint, eval, cuda, model(image, focal) -> 6 outputs -> .cpu().numpy().squeeze()."""
    import importlib
    import sys
    ckpt_dir = tmp_path / "bts_eigen_v2_pytorch_densenet161"
    ckpt_dir.mkdir()
    (ckpt_dir / "bts_eigen_v2_pytorch_densenet161.py").write_text("from bts_amd.bts import *\n")   # INTEGRATION.md, recipe A
    args = E_args = __import__("bts_amd.evaltools", fromlist=["x"]).parse_args(
        ["--encoder", "densenet161_bts", "--dataset", "kitti", "--max_depth", "80", "--input_height", "64", "--input_width", "96",
         "--model_name", "bts_eigen_v2_pytorch_densenet161", "--checkpoint_path", str(ckpt_dir / "model")])
    model_dir = os.path.dirname(args.checkpoint_path)
    sys.path.append(model_dir)
    ns = {}
    for key, val in vars(importlib.import_module(args.model_name)).items():
        if key.startswith('__') and key.endswith('__'):
            continue
        ns[key] = val
    BtsModel = ns["BtsModel"]
    assert all(k in ns for k in ("encoder", "bts", "atrous_conv", "upconv", "reduction_1x1", "local_planar_guidance",
                                 "silog_loss", "weights_init_xavier", "bn_init_as_tf"))
    # a checkpoint as bts_main.py:723-728 writes it (saved from the DataParallel/DDP wrapper)
    torch.manual_seed(4)
    src = torch.nn.DataParallel(BtsModel(params=args))
    torch.save({'global_step': 7, 'model': src.state_dict()}, args.checkpoint_path)
    model = BtsModel(params=args)
    model = torch.nn.DataParallel(model)
    checkpoint = torch.load(args.checkpoint_path, weights_only=True)
    model.load_state_dict(checkpoint['model'])
    model.eval()
    model.cuda()
    image = torch.from_numpy(synth.image_batch(1, 64, 96, 3))
    focal = torch.from_numpy(synth.focal_values(1, "kitti", 3))
    with torch.no_grad():
        lpg8x8, lpg4x4, lpg2x2, reduc1x1, depth_est, _ = model(image.cuda(), focal.cuda())
        pred = depth_est.cpu().numpy().squeeze()
        ref = src.module.eval().cuda()(image.cuda(), focal.cuda())[4].cpu().numpy().squeeze()
    assert pred.shape == (64, 96) and np.isfinite(pred).all() and (pred > 0).all()
    assert lpg8x8[0].shape == (1, 64, 96)
    assert np.array_equal(pred, ref)                          # the checkpoint round trip changes nothing
    assert model.module.decoder.lpg8x8.abs_min is not None    # bts_main.py:484-486 reads these
    sys.path.remove(model_dir)


def test_graphed_model_replay_matches_eager():
    """bts_amd.graph.GraphedModel: capture once per shape, replay for new inputs; same bits as the eager forward."""
    from bts_amd import bts as M
    from bts_amd.graph import GraphedModel
    params = Params("densenet161_bts", 512, 80.0, "kitti")
    torch.manual_seed(6)
    m = M.BtsModel(params).eval().cuda()
    gm = GraphedModel(m)
    with torch.no_grad():
        for seed in (1, 2, 3):
            x = torch.from_numpy(synth.image_batch(2, 64, 96, seed)).cuda()
            focal = torch.from_numpy(synth.focal_values(2, "kitti", seed)).cuda()
            ref = [o.clone() for o in m(x, focal)]
            got = gm(x, focal)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), "graph replay differs (seed %d)" % seed
        assert len(gm._graphs) == 1
        x2 = torch.from_numpy(synth.image_batch(1, 32, 64, 9)).cuda()          # a second shape gets its own graph
        f2 = torch.from_numpy(synth.focal_values(1, "kitti", 9)).cuda()
        assert all(torch.equal(a, b) for a, b in zip([o.clone() for o in m(x2, f2)], gm(x2, f2)))
        assert len(gm._graphs) == 2


def test_pooling_kernels():
    from bts_amd import ops
    import torch.nn.functional as F
    rng = np.random.Generator(np.random.PCG64(8))
    B, C, h, w = 2, 24, 9, 14
    x = torch.from_numpy(rng.standard_normal(size=(B, C, h, w), dtype=np.float32))
    src = torch.zeros(B * h * w, C + 8, device="cuda")
    ops.nchw_to_nhwc(x.cuda(), src[:, 4:4 + C])
    ho, wo = (h + 1) // 2, (w + 1) // 2
    d1 = torch.zeros(B * ho * wo, C, device="cuda")
    d2 = torch.zeros(B * ho * wo, C + 4, device="cuda")
    ops.maxpool3x3s2(src[:, 4:4 + C], B, h, w, d1, d2[:, 4:])
    ref = F.max_pool2d(x, 3, 2, 1)
    assert torch.equal(ops.nhwc_to_nchw(d1, B, ho, wo).cpu(), ref)
    assert torch.equal(ops.nhwc_to_nchw(d2[:, 4:], B, ho, wo).cpu(), ref)
    h, w = 8, 14
    x = torch.from_numpy(rng.standard_normal(size=(B, C, h, w), dtype=np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, size=C).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(size=C).astype(np.float32) * 0.3)
    src = torch.zeros(B * h * w, C, device="cuda")
    ops.nchw_to_nhwc(x.cuda(), src)
    dst = torch.zeros(B * (h // 2) * (w // 2), C, device="cuda")
    ops.bn_relu_avgpool2(src, B, h, w, sc.cuda(), sh.cuda(), dst)
    ref = F.avg_pool2d(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 2, 2)
    torch.testing.assert_close(ops.nhwc_to_nchw(dst, B, h // 2, w // 2).cpu(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("c,groups,stride,hw", [(128, 32, 1, (9, 13)), (256, 32, 2, (10, 14)), (512, 32, 1, (5, 7)),
                                                (1024, 32, 2, (6, 8)), (2048, 32, 1, (3, 4))])
def test_grouped_conv_bundles_and_residual_vs_torch(c, groups, stride, hw):
    """ResNeXt's grouped 3x3 as channel bundles (bts_conv_desc.n_bundles), BN+ReLU epilogue; then a 1x1 with the
    residual-add epilogue (bts_conv_desc.res) -- against F.conv2d(groups=...) / explicit add in fp64 on the CPU."""
    import torch.nn.functional as F
    from bts_amd import ops
    h, w = hw
    B = 2
    gen = torch.Generator().manual_seed(c + stride)
    x = torch.randn(B, c, h, w, generator=gen)
    wt = torch.randn(c, c // groups, 3, 3, generator=gen) / np.sqrt(9 * c // groups)
    sc, sh = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.1
    ref = F.relu(F.conv2d(x.double(), wt.double(), stride=stride, padding=1, groups=groups) * sc.double().view(1, -1, 1, 1)
                 + sh.double().view(1, -1, 1, 1))
    ho, wo = ref.shape[2:]
    wp, nb, cb = ops.pack_grouped_conv_weight(wt.cuda(), groups)
    x2d = x.cuda().permute(0, 2, 3, 1).reshape(B * h * w, c).contiguous()
    y = torch.empty(B * ho * wo, c, device="cuda")
    ops.conv_forward(x2d, B, h, w, wp, cb, 3, stride=stride, pad=1, c_in_ld=cb, e1=(sc.cuda(), sh.cuda()), act=ops.ACT_RELU,
                     y2d=y, n_bundles=nb, c_in_real=c // groups)
    got = y.view(B, ho, wo, c).permute(0, 3, 1, 2).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # 1x1 + BN + residual + ReLU
    w3 = torch.randn(64, c, 1, 1, generator=gen) / np.sqrt(c)
    res = torch.randn(B, 64, ho, wo, generator=gen)
    ref3 = F.relu(F.conv2d(ref, w3.double()) * 0.7 + 0.1 + res.double())
    w3p, co3, _ = ops.pack_conv_weight(w3.cuda())
    e3 = (torch.full((co3,), 0.7, device="cuda"), torch.full((co3,), 0.1, device="cuda"))
    res2d = res.cuda().permute(0, 2, 3, 1).reshape(B * ho * wo, 64).contiguous()
    y3 = torch.empty(B * ho * wo, 64, device="cuda")
    ops.conv_forward(y, B, ho, wo, w3p, 64, 1, e1=e3, act=ops.ACT_RELU, y2d=y3, res2d=res2d)
    got3 = y3.view(B, ho, wo, 64).permute(0, 3, 1, 2).cpu().double()
    assert (got3 - ref3).abs().max().item() <= 5e-5 * ref3.abs().max().item()


@pytest.mark.parametrize("enc,shape", [("resnet50_bts", (2, 64, 96)), ("resnext50_bts", (1, 96, 64)),
                                       ("resnext101_bts", (2, 64, 96))])
def test_resnet_hip_taps_vs_torch_cpu(enc, shape):
    """ResNet / ResNeXt encoders on the HIP conv kernel (residual epilogue, grouped bundles) vs the same nn modules
    on the CPU: the reference's tap list [x, relu, layer1..layer4] (bts.py:318-338)."""
    from bts_amd import bts as M
    from bts_amd.encoder_hip import ResNetHip
    torch.manual_seed(3)
    e = M.encoder(Params(enc, 512, 80.0, "kitti")).eval()
    _randomise_bn(e, 5)
    B, H, W = shape
    x = torch.from_numpy(synth.image_batch(B, H, W, 21))
    with torch.no_grad():
        ref = e(x)
    eg = M.encoder(Params(enc, 512, 80.0, "kitti")).eval()
    eg.load_state_dict(e.state_dict())
    eg = eg.cuda()
    plan = ResNetHip(eg.base_model)
    with torch.no_grad():
        got = plan.taps_nchw(x.cuda())
    assert len(got) == len(ref) == 6
    for i in range(1, 6):
        r, g = ref[i], got[i].cpu()
        assert r.shape == g.shape
        scale = r.abs().max().item()
        err = (r - g).abs().max().item()
        assert err <= 3e-4 * scale + 1e-5, "tap %d: max abs err %g (scale %g)" % (i, err, scale)


def test_btsmodel_resnext101_fused_forward_vs_cpu():
    """BASELINE config 3's model (ResNeXt101 plan, NYU head): BtsModel.forward all on HIP == CPU torch encoder + oracle."""
    from bts_amd import bts as M
    params = Params("resnext101_bts", 512, 10.0, "nyu")
    torch.manual_seed(12)
    model = M.BtsModel(params).eval()
    _randomise_bn(model.encoder, 8)
    feat = synth.ENCODER_CHANNELS[params.encoder]
    state_np = synth.decoder_state(feat, 512, 0)
    model.decoder.load_state_dict({k: (torch.tensor(v) if np.ndim(v) == 0 else torch.from_numpy(v.copy())) for k, v in state_np.items()})
    B, H, W = 2, 64, 96
    x = torch.from_numpy(synth.image_batch(B, H, W, 6))
    focal = torch.from_numpy(synth.focal_values(B, "nyu", 6))
    with torch.no_grad():
        ref_outs, inter = O.decoder_forward(O.state_from_numpy(state_np), model.encoder(x), focal, 10.0, "nyu",
                                            want_intermediates=True)
    mg = M.BtsModel(params).eval()
    mg.load_state_dict(model.state_dict())
    mg = mg.cuda()
    with torch.no_grad():
        got = mg(x.cuda(), focal.cuda())
    assert [type(pl).__name__ for pl in mg._enc_plans.values()] == ["ResNetHip"]
    rep = check_outputs(got, ref_outs, inter, rel_tol=5e-4, what="fused ResNeXt101 BtsModel")
    print("fused ResNeXt101 model max-rel:", rep)
