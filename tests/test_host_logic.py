"""CPU: host-side logic -- state_dict compatibility with the reference, C-ABI export list, loud failure
without a GPU, weight packing layouts."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from bts_amd import synth
from parity_util import Params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bts_hip.h")).read()
    declared = set(re.findall(r"\b(bts_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("bts_conv_desc")
    lib = ctypes.CDLL(os.path.join(ROOT, "bts_amd", "libbts_hip.so"))
    for sym in sorted(declared):
        assert hasattr(lib, sym), "include/bts_hip.h declares %s but libbts_hip.so does not export it" % sym
    from bts_amd import _lib
    assert set(_lib.SYMBOLS) == declared
    lib.bts_hip_abi_version.restype = ctypes.c_int
    assert lib.bts_hip_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define BTS_HIP_ABI_VERSION (\d+)", hdr).group(1))


def test_conv_desc_layout_matches_header():
    from bts_amd._lib import ConvDesc
    assert ctypes.sizeof(ConvDesc) == 272
    assert ConvDesc.w.offset == 56 and ConvDesc.y.offset == 136 and ConvDesc.y_nchw.offset == 152
    assert ConvDesc.tail_planes.offset == 216 and ConvDesc.n_tail.offset == 248 and ConvDesc.w_split.offset == 256 and ConvDesc.w_wino.offset == 264


@pytest.mark.parametrize("enc", ["densenet161_bts", "resnext101_bts", "densenet121_bts", "resnet50_bts"])
def test_decoder_state_dict_matches_reference_keys(enc):
    """Keys/shapes of the reference decoder (dumped from the import, see gen_golden.py strict load) ==
    synth.decoder_param_shapes == the product module's state_dict."""
    from bts_amd import bts as M
    feat = synth.ENCODER_CHANNELS[enc]
    dec = M.bts(Params(enc, 512, 80.0, "kitti"), feat, 512)
    sd = dec.state_dict()
    shapes = synth.decoder_param_shapes(feat, 512)
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    if enc == "densenet161_bts":
        assert len(sd) == 110
        assert tuple(sd["daspp_6.atrous_conv.first_bn.weight"].shape) == (576,)
        assert tuple(sd["reduc8x8.reduc.inter_128_64.0.weight"].shape) == (64, 128, 1, 1)
        assert tuple(sd["reduc1x1.reduc.final.0.weight"].shape) == (1, 8, 1, 1)
        assert tuple(sd["upconv5.conv.weight"].shape) == (512, 2208, 3, 3)


def test_btsmodel_tree_and_encoder_taps():
    """BtsModel(params) attribute tree (encoder / decoder / decoder.lpg*.abs_min) and the encoder's tap
    shapes (bts.py:327-338), on CPU (the encoder is plain torch)."""
    from bts_amd import bts as M
    m = M.BtsModel(Params("densenet161_bts", 512, 80.0, "kitti")).eval()
    assert m.decoder.lpg8x8.abs_min is None
    keys = list(m.state_dict().keys())
    assert keys[0] == "encoder.base_model.conv0.weight"
    assert "encoder.base_model.denseblock3.denselayer36.conv2.weight" in keys
    assert "encoder.base_model.transition2.norm.running_var" in keys
    assert "decoder.get_depth.0.weight" in keys
    n_params = sum(p.numel() for p in m.parameters())
    assert abs(n_params - 47.0e6) < 0.5e6          # reference README: DenseNet161 BTS 47.0 M
    with torch.no_grad():
        taps = m.encoder(torch.zeros(1, 3, 64, 96))
    assert [tuple(t.shape[1:]) for t in taps[1:]] == [(96, 32, 48), (96, 16, 24), (192, 8, 12), (384, 4, 6), (2208, 2, 3)]
    r = M.BtsModel(Params("resnext101_bts", 512, 10.0, "nyu")).eval()
    assert "encoder.base_model.layer3.22.conv2.weight" in r.state_dict()
    assert tuple(r.state_dict()["encoder.base_model.layer1.0.conv2.weight"].shape) == (256, 8, 3, 3)
    assert abs(sum(p.numel() for p in r.parameters()) - 112.8e6) < 1.5e6   # README: 112.8 M
    with torch.no_grad():
        taps = r.encoder(torch.zeros(1, 3, 64, 64))
    assert [t.shape[1] for t in taps[1:]] == [64, 256, 512, 1024, 2048]


def test_hot_path_fails_loudly_without_gpu():
    from bts_amd import bts as M, ops
    from bts_amd._lib import BtsHipError
    with pytest.raises(BtsHipError):
        ops.lpg_forward(torch.zeros(1, 4, 2, 2), 2)
    dec = M.bts(Params("densenet161_bts", 512, 80.0, "kitti"), synth.ENCODER_CHANNELS["densenet161_bts"], 512).eval()
    feats = [None] + [torch.from_numpy(f) for f in synth.encoder_features(synth.ENCODER_CHANNELS["densenet161_bts"], 1, 32, 32)[1:]]
    with pytest.raises(BtsHipError):
        dec(feats, torch.ones(1))


def test_pack_layouts():
    from bts_amd import ops
    w = torch.arange(2 * 5 * 9, dtype=torch.float32).reshape(2, 5, 3, 3)
    p, cop, cld = ops.pack_conv_weight(w)                  # K flattened tap-major: k = tap*c_in_ld + c
    assert (cop, cld) == (32, 8) and tuple(p.shape) == (32, 96)      # 9*8 = 72 -> 96
    assert p[1, 4 * 8 + 3] == w[1, 3, 1, 1] and p[0, 0] == w[0, 0, 0, 0] and p[1, 8 * 8 + 4] == w[1, 4, 2, 2]
    assert p[2:].abs().sum() == 0 and p[:, 72:].abs().sum() == 0 and p[:, 5:8].abs().sum() == 0
    perm = torch.tensor([4, 0, 1, 2, 3])
    pp, _, _ = ops.pack_conv_weight(w, perm=perm)
    assert pp[1, 4 * 8 + 0] == w[1, 4, 1, 1]
    # wide-chain fragments (32x32x2 MFMA): float4 ((mt*(K/8)+g)*64 + 32h + i) = W[32mt+i][4(2g+h) .. +3]
    ww = torch.arange(64 * 128, dtype=torch.float32).reshape(64, 128, 1, 1)
    f = ops.pack_reduc_weights([ww]).view(-1, 4)
    K = 128
    for (mt, g, h, i) in ((0, 0, 0, 0), (1, 5, 1, 7), (0, 15, 0, 31), (1, 2, 1, 20)):
        got = f[(mt * (K // 8) + g) * 64 + 32 * h + i]
        assert torch.equal(got, ww[32 * mt + i, 4 * (2 * g + h):4 * (2 * g + h) + 4, 0, 0])
    # narrow-chain fragments (16x16x4 MFMA): float4 ((mt*G+g)*64 + 16kq + i) = W[16mt+i][16g + 4kq .. +3], zero padded
    w1 = torch.arange(16 * 32, dtype=torch.float32).reshape(16, 32, 1, 1) + 1
    w2 = torch.arange(8 * 16, dtype=torch.float32).reshape(8, 16, 1, 1) + 1
    w3 = torch.arange(1 * 8, dtype=torch.float32).reshape(1, 8, 1, 1) + 1
    assert ops.reduc_uses_mfma16(32, 16) and not ops.reduc_uses_mfma16(128, 64)
    f = ops.pack_reduc_weights([w1, w2, w3]).view(-1, 4)
    assert f.shape[0] == (2 + 1 + 1) * 64
    for (g, kq, i) in ((0, 0, 0), (1, 3, 15), (0, 2, 9)):
        assert torch.equal(f[g * 64 + 16 * kq + i], w1[i, 16 * g + 4 * kq:16 * g + 4 * kq + 4, 0, 0])
    l2, l3 = f[128:192], f[192:256]
    assert torch.equal(l2[16 * 1 + 3], w2[3, 4:8, 0, 0]) and l2[16 * 1 + 9].abs().sum() == 0      # rows 8..15 are padding
    assert torch.equal(l3[16 * 1 + 0], w3[0, 4:8, 0, 0]) and l3[16 * 2 + 0].abs().sum() == 0      # k 8..15 are padding
    assert ops.reduc_chain(128, 128) == [(128, 128), (128, 64), (64, 32), (32, 16), (16, 8), (8, -1)]
    assert synth.reduc_chain_channels(32, 16, True) == [32, 16, 8, 1]


def test_trainer_protocol_pieces():
    """bts_main.py's model-side training protocol: poly LR decay (:420,:603), frozen-layer fragments (:165-182),
    AdamW groups (:354-356), ground-truth mask thresholds (:551-553)."""
    from collections import namedtuple
    from bts_amd import bts as M, trainer
    assert trainer.poly_lr(0, 1000, 1e-4) == pytest.approx(1e-4)
    assert trainer.poly_lr(1000, 1000, 1e-4) == pytest.approx(1e-5)
    assert trainer.poly_lr(500, 1000, 1e-4, 2e-5) == pytest.approx((1e-4 - 2e-5) * 0.5 ** 0.9 + 2e-5)
    assert trainer.fixing_layers("densenet161_bts") == ['conv0', 'norm']
    assert trainer.fixing_layers("resnext101_bts", fix_first_conv_blocks=True) == \
        ['base_model.conv1', 'base_model.layer1.0', 'base_model.layer1.1', '.bn']
    assert trainer.fixing_layers("densenet121_bts", fix_first_conv_block=True) == ['conv0', 'denseblock1.denselayer1', 'norm']
    P = namedtuple("P", "encoder bts_size max_depth dataset")
    model = M.BtsModel(P("densenet121_bts", 512, 80.0, "kitti"))
    frozen = trainer.set_misc(model, "densenet121_bts")
    names = dict(model.encoder.named_parameters())
    assert "base_model.conv0.weight" in frozen and not names["base_model.conv0.weight"].requires_grad
    assert all(("norm" in n or "conv0" in n) for n in frozen)
    assert names["base_model.denseblock1.denselayer1.conv1.weight"].requires_grad
    assert all(p.requires_grad for p in model.decoder.parameters())
    opt = trainer.make_optimizer(model, 1e-4, 1e-2, 1e-3)
    assert [g["weight_decay"] for g in opt.param_groups] == [1e-2, 0]
    assert opt.param_groups[0]["eps"] == 1e-3
    gt = torch.tensor([0.05, 0.5, 1.5])
    assert trainer.gt_mask(gt, "kitti").tolist() == [False, False, True]
    assert trainer.gt_mask(gt, "nyu").tolist() == [False, True, True]


def test_caches_survive_deepcopy_pickle_and_replication():
    """The packed-weight / workspace caches (bts_amd/workspace.py) must not break what callers do with the model
    object: copy.deepcopy (tests, EMA copies), torch.save(model) (a lock is not picklable), and DataParallel's
    replicate(), whose shallow __dict__ copy is how replicas find the SOURCE module's caches (bts_test.py:91)."""
    import copy
    import io
    from types import SimpleNamespace
    from bts_amd import bts as M
    from bts_amd.encoder_hip import DenseNetHip
    from bts_amd.workspace import WorkspaceCache
    m = M.BtsModel(SimpleNamespace(encoder="densenet121_bts", bts_size=512, max_depth=80.0, dataset="kitti"))
    m._enc_plans["cuda:0"] = DenseNetHip(m.encoder.base_model, key_module=m.encoder.base_model)
    c = copy.deepcopy(m)
    assert c._origin[0] is c and c.decoder._packs.origin is c.decoder and c.decoder._bufs is not m.decoder._bufs
    assert c._enc_plans["cuda:0"].features is c.encoder.base_model
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)            # a file this test wrote itself
    assert r._origin[0] is r and r.decoder.daspp_6._packs.origin is r.decoder.daspp_6
    rep = m.decoder._replicate_for_data_parallel()
    assert rep._packs is m.decoder._packs and rep._bufs is m.decoder._bufs and rep._packs.origin is m.decoder
    # LRU eviction over UNPINNED entries only: pinned ones (live graphs) are extra and never dropped
    wc = WorkspaceCache(max_entries=2)
    for k in "abc":
        wc.get(k, dict)
        wc.pin(k)
    wc.get("d", dict)
    wc.get("e", dict)
    assert len(wc) == 5 and all(k in wc for k in "abcde")        # 3 pinned + 2 unpinned
    wc.get("f", dict)
    assert "d" not in wc and all(k in wc for k in "abcef")      # oldest unpinned one went
    wc.unpin("a")
    assert "a" not in wc and len(wc) == 4                        # unpinned and least recently used: a goes, e and f stay
    wc.get("g", dict)
    assert all(k in wc for k in "bcfg") and len(wc) == 4


def test_plan_op_layout_matches_header(tmp_path):
    """bts_amd/plan.py mirrors `bts_op` / `bts_plan_patch` (include/bts_hip.h) with ctypes structures derived from the
    declared argtypes: sizes and the union offset must equal what a C compiler makes of the header."""
    import subprocess
    from bts_amd import plan
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bts_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(bts_op), offsetof(bts_op, u), sizeof(bts_plan_patch), sizeof(bts_conv_desc), '
                   'offsetof(bts_op, u.reduc_lpg.abs_min));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    c_sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    plan._build_types()
    rl = plan._structs["bts_reduc_lpg_fwd_f32"]
    py_sizes = [ctypes.sizeof(plan._BtsOp), plan._BtsOp.u.offset, ctypes.sizeof(plan._BtsPatch),
                ctypes.sizeof(plan._structs["bts_conv_fwd_f32"]), plan._BtsOp.u.offset + getattr(rl, "a14").offset]
    assert c_sizes == py_sizes, (c_sizes, py_sizes)


# ------------------------------------------------------------------------------------------ tap skipping (host side)
def _ksteps(B, h, w, cin, cout, k, dil, pad, fill=0):
    """issued / dense tap-steps of a conv launch, through the library's host-side query (no GPU work; the pointers are
    never dereferenced)."""
    import ctypes as C
    from bts_amd import _lib
    d = _lib.ConvDesc()
    d.x = d.w = d.y = 0x1000
    d.x_pix_stride = cin
    d.c_in_ld = cin
    d.k_pad = (k * k * cin + 31) // 32 * 32
    d.B, d.h_in, d.w_in, d.up = B, h, w, 1
    d.ksize, d.dil, d.stride, d.pad = k, dil, 1, pad
    d.c_out, d.c_out_pad = cout, (cout + 31) // 32 * 32
    d.y_pix_stride = cout
    d.fill_frames = fill
    issued, dense = C.c_long(-1), C.c_long(-1)
    bm, bn, kind = C.c_int(0), C.c_int(0), C.c_int(0)
    lib = _lib.load_real()
    assert lib.bts_conv_plan_f32(C.byref(d), C.byref(bm), C.byref(bn), C.byref(kind)) == 0
    assert lib.bts_conv_plan_ksteps_f32(C.byref(d), C.byref(issued), C.byref(dense)) == 0
    return issued.value, dense.value, bm.value, kind.value


@pytest.mark.parametrize("B,h,w,dil", [(2, 44, 152, 24), (2, 44, 152, 18), (1, 44, 152, 6), (3, 52, 68, 24), (1, 13, 17, 24), (2, 11, 19, 6), (1, 52, 68, 3)])      # (dilation 3 on 44x152 takes the dilated halo tile)
def test_tap_skipping_rule_is_a_superset_of_the_taps_a_tile_needs(B, h, w, dil):
    """tile_tapmask (conv_mfma.hip) may only drop a tap that lies in the zero padding for EVERY pixel of the row tile
    (reference: the dilated 3x3 of atrous_conv, bts.py:75-77, padding = dilation).  Brute force over all pixels: the
    issued count must cover every (tile, tap) pair with at least one in-map read, and stay below the dense count
    where the dilation exceeds half the map."""
    issued, dense, bm, kind = _ksteps(B, h, w, 256, 128, 3, dil, dil)
    assert kind == 0 and bm in (64, 128)                    # row-tiled kernel, no split-K
    M = B * h * w
    n_mt = (M + bm - 1) // bm
    assert dense == n_mt * 9
    m = np.arange(M)
    y, x = (m % (h * w)) // w, m % w
    needed = 0
    for ky in range(3):
        for kx in range(3):
            ok = (y + (ky - 1) * dil >= 0) & (y + (ky - 1) * dil < h) & (x + (kx - 1) * dil >= 0) & (x + (kx - 1) * dil < w)
            needed += int(np.add.reduceat(ok, np.arange(0, M, bm)).astype(bool).sum())
    assert needed <= issued <= dense
    if 2 * dil > h:                                          # e.g. dilation 24 on 44 rows: most tiles lose a kernel row
        assert issued < 0.8 * dense
    assert issued - needed <= 0.1 * dense                    # the first/last-pixel rule is close to exact


def test_tap_skipping_off_where_it_cannot_apply():
    """1x1 convolutions and split-K launches report no skipping (issued == dense or both 0)."""
    issued, dense, _, kind = _ksteps(2, 44, 152, 256, 256, 1, 1, 0)
    assert issued == dense


# ------------------------------------------------------------------------------------------ dispatch vs declared fill (host side)
def _plan(B, h, w, cin, cout, k, fill, ws_floats=1 << 28):
    """(kernel kind, bm, bn) bts_conv_fwd_f32 would pick -- host-side query, pointers never dereferenced."""
    import ctypes as C
    from bts_amd import _lib
    d = _lib.ConvDesc()
    d.x = d.w = d.y = 0x1000
    d.splitk_ws, d.splitk_ws_floats = 0x2000, ws_floats
    d.x_pix_stride = cin
    d.c_in_ld = cin
    d.k_pad = (k * k * cin + 31) // 32 * 32
    d.B, d.h_in, d.w_in, d.up = B, h, w, 1
    d.ksize, d.dil, d.stride, d.pad = k, 1, 1, k // 2
    d.c_out, d.c_out_pad = cout, (cout + 31) // 32 * 32
    d.y_pix_stride = cout
    d.fill_frames = fill
    bm, bn, kind = C.c_int(0), C.c_int(0), C.c_int(0)
    assert _lib.load_real().bts_conv_plan_f32(C.byref(d), C.byref(bm), C.byref(bn), C.byref(kind)) == 0
    return kind.value, bm.value, bn.value


def test_block3_kernel_choices_need_a_declaration_above_the_default():
    """DenseNet block 3 (22x76 maps): the wide 1x1 tile and the halo kernel instead of split-K are only worth it when
    enough frames share the launch.  The library default (fill_frames 0 -> 8) and a single-frame caller must stay on
    the split-K / row-tiled kernels -- with the block-3 choices at the default, batch 1 cost 12.0 instead of 8.3 ms per
    frame -- while a caller that declares 16 frames (bench.py at B=16) gets them.  Never a function of B."""
    for B in (1, 16):
        for fill in (0, 1, 2, 8):
            kind, bm, bn = _plan(B, 22, 76, 1248, 192, 1, fill)
            assert kind & 15 == 0, (B, fill, kind)                 # row-tiled kernel
            kind, _, bn = _plan(B, 22, 76, 192, 48, 3, fill)
            assert kind & 15 == 0 and kind & 16 and bn == 48, (B, fill, kind)      # row-tiled + split-K
        kind, bm, bn = _plan(B, 22, 76, 1248, 192, 1, 16)
        assert kind & 15 == 3 and (bm, bn) == (128, 192), (B, kind, bm, bn)       # wide 1x1, 128-row tile (K > 768)
        kind, bm, bn = _plan(B, 22, 76, 432, 192, 1, 16)
        assert kind & 15 == 3 and (bm, bn) == (64, 192)                             # K <= 768: 64-row tile
        kind, _, bn = _plan(B, 22, 76, 192, 48, 3, 16)
        assert kind & 15 == 1 and kind & 32 and not kind & 16 and bn == 48         # halo kernel, eight-wave 48-wide tile
    # blocks 1-2 fill the chip at any declaration: wide 1x1 / halo kernels whatever the fill
    for fill in (0, 1, 16):
        assert _plan(1, 88, 304, 240, 192, 1, fill)[0] & 15 == 3
        assert _plan(1, 88, 304, 192, 48, 3, fill)[0] & 15 == 1


def test_launch_config_is_a_thread_local_scope():
    """ops.launch_config: the (fill_frames, precision) declaration of a forward is scoped to the calling thread and
    nests; outside any scope the library defaults apply.  A model's default (fill_frames None) follows the batch of
    the call in three classes: single frames (bts_test.py's loop) -> 2, small batches -> the library default 8,
    chip-filling batches -> 16."""
    import threading
    from types import SimpleNamespace
    from bts_amd import ops
    assert ops.current_launch_config() == (0, 0) and not ops.launch_config_active()
    seen = {}

    def other():
        seen["other"] = (ops.current_launch_config(), ops.launch_config_active())

    with ops.launch_config(fill_frames=2, precision="bf16x3"):
        assert ops.current_launch_config() == (2, 1) and ops.launch_config_active()
        th = threading.Thread(target=other)
        th.start()
        th.join()
        with ops.launch_config(precision="fp32"):
            assert ops.current_launch_config() == (2, 0)
        assert ops.current_launch_config() == (2, 1)
    assert seen["other"] == ((0, 0), False)
    assert ops.current_launch_config() == (0, 0) and not ops.launch_config_active()
    assert [ops.auto_fill_frames(b) for b in (1, 2, 3, 4, 8, 11, 12, 16, 64)] == [2, 2, 8, 8, 8, 8, 16, 16, 16]
    with ops.model_launch_config(SimpleNamespace(fill_frames=None, conv_precision="fp32"), 1):
        assert ops.current_launch_config() == (2, 0)
    with ops.model_launch_config(SimpleNamespace(fill_frames=5, conv_precision="bf16x3"), 16):
        assert ops.current_launch_config() == (5, 1)
    for bad in (dict(fill_frames=-1), dict(fill_frames=5000), dict(precision="fp16")):
        with pytest.raises(ops.BtsHipError):
            ops.launch_config(**bad)
    assert not hasattr(ops, "set_fill_frames") and not hasattr(ops, "set_conv_precision")


def test_torch_operator_library_registers_the_hot_path_ops():
    """libbts_torch.so (csrc/torch_ops.cpp): TORCH_LIBRARY(bts_hip) over the C ABI -- the operator boundary north_star
    names.  On a machine without a GPU: the library loads, every operator is registered with the schema the Python side
    calls, and a CPU tensor is refused loudly (there is no CPU kernel to fall back to)."""
    import torch
    from bts_amd import _lib, ops
    t = _lib.load_torch_ops()
    want = {"lpg": "bts_hip::lpg(Tensor plane_eq, int upratio) -> (Tensor, Tensor)",
            "lpg_backward": "bts_hip::lpg_backward(Tensor plane_eq, Tensor grad_depth, int upratio) -> Tensor",
            "reduction_1x1": None, "reduc_lpg": None, "conv_fwd": None}
    for name, schema in want.items():
        op = getattr(t, name)
        s = str(op.default._schema)
        assert s.startswith("bts_hip::" + name + "("), s
        if schema:
            assert s == schema, s
    assert "Tensor(a!) y" in str(t.conv_fwd.default._schema) and "int[] geom" in str(t.conv_fwd.default._schema)
    with pytest.raises((RuntimeError, NotImplementedError)):
        t.lpg(torch.zeros(1, 4, 2, 2), 8)                    # no CPU backend registered
    assert ops.torch_ops() is t                               # the default binding of the hot-path operators
    with pytest.raises(ops.BtsHipError):
        ops.lpg_forward(torch.zeros(1, 4, 2, 2), 8)


def test_pack_wino_size_query_on_the_host():
    """bts_pack_wino_floats is pure host arithmetic (no GPU): 16 transform positions x buffer channels x output rows; the
    planar-tail form leaves the last four channels out; shapes the fused Winograd kernel cannot take return -1."""
    from bts_amd import _lib
    lib = _lib.load()
    assert lib.bts_pack_wino_floats(256, 448, 0, 0) == 16 * 448 * 256            # conv4: 32-wide channel tiles
    assert lib.bts_pack_wino_floats(64, 192, 0, 48) == 16 * 192 * 48             # DenseNet growth conv: 16-wide tiles, 48 real outputs
    assert lib.bts_pack_wino_floats(128, 228, 1, 0) == 16 * 224 * 128            # conv3: 224 buffer channels + planar tail
    for bad in ((256, 450, 0, 0), (250, 448, 0, 0), (64, 192, 0, 40), (64, 192, 0, 80), (64, 36, 0, 0), (0, 64, 0, 0)):
        assert lib.bts_pack_wino_floats(*bad) == -1, bad


@pytest.mark.parametrize("h,w,cout,expect_wino", [(44, 152, 256, True), (52, 68, 256, True), (88, 304, 48, True),
                                                  (26, 34, 256, False), (11, 38, 48, False)])
def test_winograd_is_chosen_by_per_frame_geometry_only(h, w, cout, expect_wino):
    """The fused Winograd kernel takes a stride-1 3x3 layer when the caller supplies transformed weights and the map fills at
    least 0.70 of its 8x16-pixel tile grid (44x152: 0.87, the NYU decoder's 52x68: 0.79; 26x34: 0.58 and 11x38: 0.54 stay on
    the direct kernels) -- and the answer is the same for every batch size (a frame's bits may not depend on its
    neighbours).  Host-side query: no GPU work, the pointers are never dereferenced."""
    import ctypes as C
    from bts_amd import _lib
    lib = _lib.load_real()
    kinds = set()
    for B in (1, 4, 16):
        d = _lib.ConvDesc()
        d.x = d.w = d.y = d.w_wino = 0x1000
        cin = 192
        d.x_pix_stride = d.c_in_ld = cin
        d.k_pad = 9 * cin
        d.B, d.h_in, d.w_in, d.up = B, h, w, 1
        d.ksize, d.dil, d.stride, d.pad = 3, 1, 1, 1
        d.c_out, d.c_out_pad = cout, (cout + 31) // 32 * 32
        d.y_pix_stride = cout
        d.fill_frames = 16
        bm, bn, kind = C.c_int(0), C.c_int(0), C.c_int(0)
        assert lib.bts_conv_plan_f32(C.byref(d), C.byref(bm), C.byref(bn), C.byref(kind)) == 0
        kinds.add(kind.value & 15)
    assert len(kinds) == 1, kinds
    assert (kinds == {6}) == expect_wino, kinds
