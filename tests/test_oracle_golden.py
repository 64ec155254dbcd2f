"""CPU: the oracle (oracle/bts_oracle.py) must reproduce the reference-generated
goldens (tests/golden/*.npz, made by tests/golden/gen_golden.py) bit-exactly."""
import os

import numpy as np
import pytest
import torch

from bts_amd import synth
from oracle import bts_oracle as O

CONFIGS = {
    "K": ("densenet161_bts", 80.0, "kitti", 352, 1216),
    "N": ("resnext101_bts", 10.0, "nyu", 416, 544),
}
OUT_NAMES = ("depth_8x8_scaled", "depth_4x4_scaled", "depth_2x2_scaled", "reduc1x1", "final_depth", "iconv1")
MOD_SHAPES = [(2, 11, 19), (1, 13, 17)]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _eq(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape
    assert np.array_equal(a, b, equal_nan=True), "max abs diff %g" % np.nanmax(np.abs(a - b))


@pytest.fixture(scope="module")
def states():
    return {c: O.state_from_numpy(synth.decoder_state(synth.ENCODER_CHANNELS[v[0]], 512, 0))
            for c, v in CONFIGS.items()}


@pytest.mark.parametrize("k", [2, 4, 8])
def test_lpg_tables(golden_dir, k):
    g = np.load(os.path.join(golden_dir, "lpg_tables.npz"))
    for name in ("hand", "orient", "rand"):
        y, am = O.lpg_forward(_t(g["%s_%d_in" % (name, k)]), k)
        _eq(y.numpy(), g["%s_%d_out" % (name, k)])
        assert np.float32(am.item()) == g["%s_%d_absmin" % (name, k)]
    # the clamp table itself: den {5e-4,-5e-4,0,2e-3} with n4=0.5
    hand = g["hand_%d_out" % k][0, 0]
    cells = hand.reshape(-1, k)[:, 0]
    np.testing.assert_allclose(cells[:2], [500.0, -500.0], rtol=1e-6)
    assert np.isinf(cells[2]) and cells[2] > 0
    np.testing.assert_allclose(cells[3:5], [250.0, -250.0], rtol=1e-6)


def test_lpg_orientation(golden_dir):
    """u varies along columns, v along rows (bts.py:142-143)."""
    g = np.load(os.path.join(golden_dir, "lpg_tables.npz"))
    y = g["orient_4_out"]
    assert not np.allclose(y[0, 0, :4], y[0, 0, 0])       # n1 case: changes along a row's columns
    assert np.allclose(y[0, :4, 0], y[0, 0, 0])           # ... constant down a column
    assert np.allclose(y[1, 0, :4], y[1, 0, 0])           # n2 case: the opposite
    assert not np.allclose(y[1, :4, 0], y[1, 0, 0])


@pytest.mark.parametrize("cname", ["K", "N"])
def test_reduction_modules(golden_dir, states, cname):
    g = np.load(os.path.join(golden_dir, "modules_small.npz"))
    p = states[cname]
    reducs = {"reduc8x8": (128, False), "reduc4x4": (128, False), "reduc2x2": (64, False), "reduc1x1": (32, True)}
    for name, (cin, fin) in reducs.items():
        ws = O._reduc_weights(p, name)
        for md in (80.0, 10.0):
            for si, (B, h, w) in enumerate(MOD_SHAPES):
                rng = np.random.Generator(np.random.PCG64(2000 + si))
                x = rng.standard_normal(size=(B, cin, h, w), dtype=np.float32)
                y = O.reduction_forward(_t(x), ws, md, fin)
                _eq(y.numpy(), g["%s_%s_md%d_s%d" % (cname, name, int(md), si)])


@pytest.mark.parametrize("cname", ["K", "N"])
def test_atrous_modules(golden_dir, states, cname):
    g = np.load(os.path.join(golden_dir, "modules_small.npz"))
    p = states[cname]
    feat = synth.ENCODER_CHANNELS[CONFIGS[cname][0]]
    aspp = {"daspp_3": (256, 3, False), "daspp_6": (384 + feat[2], 6, True), "daspp_12": (512 + feat[2], 12, True),
            "daspp_18": (640 + feat[2], 18, True), "daspp_24": (768 + feat[2], 24, True)}
    for name, (cin, dil, fbn) in aspp.items():
        for si, (B, h, w) in enumerate(MOD_SHAPES):
            rng = np.random.Generator(np.random.PCG64(3000 + si))
            x = rng.standard_normal(size=(B, cin, h, w), dtype=np.float32)
            y = O.atrous_forward(_t(x), p, name, dil, fbn)
            _eq(y.numpy(), g["%s_%s_s%d" % (cname, name, si)])


def _run(cname, states, B, H, W, seed):
    enc, md, ds, _, _ = CONFIGS[cname]
    feat = synth.ENCODER_CHANNELS[enc]
    feats = synth.encoder_features(feat, B, H, W, seed=seed)
    focal = synth.focal_values(B, ds, seed=seed)
    with torch.no_grad():
        return O.decoder_forward(states[cname], [None] + [_t(f) for f in feats[1:]], _t(focal), md, ds,
                                 want_intermediates=True)


@pytest.mark.parametrize("cname", ["K", "N"])
def test_decoder_small(golden_dir, states, cname):
    g = np.load(os.path.join(golden_dir, "decoder_small.npz"))
    outs, inter = _run(cname, states, 2, 64, 96, 4321)
    for n, o in zip(OUT_NAMES, outs):
        _eq(o.numpy(), g["%s_%s" % (cname, n)])
    am = np.asarray([inter["abs_min_8x8"].item(), inter["abs_min_4x4"].item(), inter["abs_min_2x2"].item()],
                    dtype=np.float32)
    _eq(am, g["%s_abs_min" % cname])


@pytest.mark.parametrize("cname", ["K", "N"])
def test_decoder_full_samples(golden_dir, states, cname):
    g = np.load(os.path.join(golden_dir, "decoder_full_samples.npz"))
    _, _, _, H, W = CONFIGS[cname]
    outs, inter = _run(cname, states, 1, H, W, 1234)
    for n, o in zip(OUT_NAMES, outs):
        flat = o.numpy().reshape(-1)
        _eq(flat[g["%s_%s_idx" % (cname, n)]], g["%s_%s_val" % (cname, n)])
        st = g["%s_%s_stats" % (cname, n)]
        fin = flat[np.isfinite(flat)]
        assert fin.min() == st[0] and fin.max() == st[1]


@pytest.mark.parametrize("k", [2, 4, 8])
def test_c_oracle_lpg_tables(golden_dir, k):
    """oracle/lpg_oracle.c (built by __graft_entry__.build()) against the same reference-generated tables."""
    import ctypes
    so = os.path.join(os.path.dirname(golden_dir), "..", "oracle", "_build", "liblpg_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(golden_dir), "..", "oracle")])
    lib = ctypes.CDLL(os.path.abspath(so))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.lpg_oracle_fwd.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp]
    g = np.load(os.path.join(golden_dir, "lpg_tables.npz"))
    for name in ("hand", "orient", "rand"):
        x = np.ascontiguousarray(g["%s_%d_in" % (name, k)])
        B, _, h, w = x.shape
        out = np.empty((B, h * k, w * k), dtype=np.float32)
        am = np.zeros(1, dtype=np.float32)
        lib.lpg_oracle_fwd(x.ctypes.data_as(fp), B, h, w, k, out.ctypes.data_as(fp), am.ctypes.data_as(fp))
        _eq(out, g["%s_%d_out" % (name, k)])
        assert am[0] == g["%s_%d_absmin" % (name, k)]


def test_decoder_train_step(golden_dir):
    """Training step (batch-stat BN, silog loss, backward) of the oracle vs the reference's own, recorded in
    decoder_train.npz by tests/golden/gen_golden.py (reference bts.py in train() mode + bts.silog_loss)."""
    from parity_util import check_train_against_golden, oracle_train_step
    g = np.load(os.path.join(golden_dir, "decoder_train.npz"))
    r = oracle_train_step()
    bufs = {k: v for k, v in r["state"].items() if k.endswith(("running_mean", "running_var"))}
    check_train_against_golden(g, r["loss"], [o.numpy() for o in r["outs"]], [f.numpy() for f in r["feat_grads"]],
                               {k: v.numpy() for k, v in r["param_grads"].items()},
                               {k: v.detach().numpy() for k, v in bufs.items()}, rtol_grad=1e-4, what="oracle")
