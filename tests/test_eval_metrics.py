"""Evaluation path: crop rectangles, prediction clean-up and the nine measures.

CPU tests pin the oracle (oracle/eval_oracle.py, the reference's code restated in its float32 NumPy arithmetic) and the
host helpers of bts_amd.evaltools against HAND-DERIVED values (tests/golden/eval_fixture.json) -- the reference's own
functions cannot be imported here (cv2 / c3d at module level) and it ships no evaluation fixtures, so this path is
"parity unpinned" by reference outputs.  GPU tests hold bts_eval_depth_metrics_f32 to the oracle."""
import json
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from bts_amd import evaltools as E
from oracle import eval_oracle as EO

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_fixture.json")))
NAMES = E.EVAL_METRICS


def test_crop_rectangles_and_kb_offsets_match_hand_derived_values():
    for gh, gw, ds, garg, eigen, y0, y1, x0, x1 in FIX["crop_rects"]:
        assert E.eval_crop_rect(gh, gw, ds, garg, eigen) == (y0, y1, x0, x1)
        # the oracle's mask (reference code) covers exactly that rectangle
        gt = np.full((gh, gw), 5.0, dtype=np.float32)
        _, n = EO.eval_sample(gt.copy(), gt, ds, 1e-3, 80.0, garg_crop=garg, eigen_crop=eigen)
        assert n == (y1 - y0) * (x1 - x0)
        _, valid = E.prepare_eval(gt.copy(), gt, ds, 1e-3, 80.0, garg_crop=garg, eigen_crop=eigen)
        assert valid.sum() == n and valid[y0:y1, x0:x1].all()
    for gh, gw, top, left in FIX["kb_crop_offsets"]:
        assert E.kb_crop_offsets(gh, gw) == (top, left)


def test_oracle_and_host_metrics_match_hand_computed_sample():
    s = FIX["sample"]
    gt = np.asarray(s["gt"], dtype=np.float32)
    pred = np.asarray(s["pred"], dtype=np.float32)
    want = [s["measures"][n] for n in NAMES]
    got, n = EO.eval_sample(pred, gt, "kitti", s["min_depth_eval"], s["max_depth_eval"])
    assert n == s["valid"]
    np.testing.assert_allclose(got, want, rtol=2e-6)                    # float32 arithmetic, as the reference
    p2, valid = E.prepare_eval(pred, gt, "kitti", s["min_depth_eval"], s["max_depth_eval"])
    np.testing.assert_allclose(E.compute_errors(gt[valid], p2[valid]), want, rtol=1e-12)     # fp64 host version
    cin = np.asarray([[float(v) for v in s["cleanup_in"]]], dtype=np.float32)
    cleaned, _ = E.prepare_eval(cin, np.ones_like(cin), "kitti", s["min_depth_eval"], s["max_depth_eval"])
    np.testing.assert_allclose(cleaned[0], np.asarray(s["cleanup_out"], dtype=np.float32))


def _random_case(rng, B, Hg, Wg, Hp, Wp, dmax):
    gt = rng.uniform(0.0, 1.2 * dmax, size=(B, Hg, Wg)).astype(np.float32)
    gt[rng.uniform(size=gt.shape) < 0.3] = 0.0                          # lidar holes
    pred = (gt[:, :Hp, :Wp] * rng.uniform(0.5, 1.6, size=(B, Hp, Wp)) + rng.uniform(0, 2, size=(B, Hp, Wp))).astype(np.float32)
    bad = rng.uniform(size=pred.shape)
    pred[bad < 0.01] = np.nan
    pred[(bad >= 0.01) & (bad < 0.02)] = np.inf
    pred[(bad >= 0.02) & (bad < 0.03)] = -np.inf
    pred[(bad >= 0.03) & (bad < 0.04)] = 0.0
    return gt, pred


CASES = [  # (B, Hg, Wg, Hp, Wp, dataset, dmax, kb, garg, eigen)
    (3, 375, 1242, 352, 1216, "kitti", 80.0, True, True, False),
    (2, 370, 1226, 352, 1216, "kitti", 80.0, True, False, True),
    (2, 352, 1216, 352, 1216, "kitti", 80.0, False, False, False),
    (4, 480, 640, 480, 640, "nyu", 10.0, False, False, True),
    (1, 5, 7, 5, 7, "nyu", 10.0, False, False, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d-%s%s%s" % (c[0], c[1], c[2], c[5], "-garg" if c[8] else "", "-eigen" if c[9] else ""))
def test_gpu_metrics_vs_oracle(case):
    B, Hg, Wg, Hp, Wp, ds, dmax, kb, garg, eigen = case
    rng = np.random.default_rng(B * 1000 + Hg)
    gt, pred = _random_case(rng, B, Hg, Wg, Hp, Wp, dmax)
    acc = torch.zeros(10, dtype=torch.float64, device="cuda")
    out = E.gpu_compute_errors(torch.from_numpy(pred).cuda().unsqueeze(1), torch.from_numpy(gt).cuda().unsqueeze(1), ds, 1e-3,
                               dmax, do_kb_crop=kb, garg_crop=garg, eigen_crop=eigen, accum=acc).cpu().numpy()
    want_acc = np.zeros(10)
    for b in range(B):
        p2, valid = E.prepare_eval(pred[b], gt[b], ds, 1e-3, dmax, do_kb_crop=kb, garg_crop=garg, eigen_crop=eigen)
        want64 = E.compute_errors(gt[b][valid], p2[valid])                       # fp64 statement of the same formulas
        want32, n = EO.eval_sample(pred[b], gt[b], ds, 1e-3, dmax, do_kb_crop=kb, garg_crop=garg, eigen_crop=eigen)
        assert out[b, 9] == n == valid.sum()
        np.testing.assert_allclose(out[b, :9], want64, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(out[b, :9], np.asarray(want32, dtype=np.float64), rtol=5e-4)   # the reference's float32 sums
        want_acc[:9] += want64
        want_acc[9] += 1
    np.testing.assert_allclose(acc.cpu().numpy(), want_acc, rtol=1e-11)
    # bit-reproducible: fixed-order fp64 sums, no atomics
    out2 = E.gpu_compute_errors(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), ds, 1e-3, dmax, do_kb_crop=kb,
                                garg_crop=garg, eigen_crop=eigen).cpu().numpy()
    assert np.array_equal(out, out2)


@pytest.mark.gpu
def test_gpu_metrics_hand_sample_and_empty_frames():
    s = FIX["sample"]
    gt = torch.tensor([s["gt"], [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]], dtype=torch.float32).cuda()
    pred = torch.tensor([s["pred"], s["pred"]], dtype=torch.float32).cuda()
    acc = torch.zeros(10, dtype=torch.float64, device="cuda")
    out = E.gpu_compute_errors(pred, gt, "kitti", s["min_depth_eval"], s["max_depth_eval"], accum=acc).cpu().numpy()
    np.testing.assert_allclose(out[0, :9], [s["measures"][n] for n in NAMES], rtol=1e-12)
    assert out[0, 9] == 4 and out[1, 9] == 0 and not out[1, :9].any()
    assert acc[9].item() == 1.0                    # a frame whose mask leaves no pixel is left out of the accumulator
    # constant scale error (pred = 1.1 gt everywhere) and a single valid pixel: every log error is the same number, the silog
    # variance E[e^2] - E[e]^2 is zero up to rounding -- it may come out a few ulp NEGATIVE: the result must be ~0, never NaN,
    # and the accumulator must stay finite
    g = torch.Generator().manual_seed(3)
    gt2 = (torch.rand((3, 40, 60), generator=g) * 60 + 2).float()
    gt2[2] = 0.0
    gt2[2, 7, 9] = 17.5
    pred2 = gt2 * 1.1
    pred2[2] = 21.25
    acc2 = torch.zeros(10, dtype=torch.float64, device="cuda")
    out2 = E.gpu_compute_errors(pred2.cuda(), gt2.cuda(), "kitti", 1e-3, 80.0, accum=acc2).cpu().numpy()
    assert np.isfinite(out2).all() and np.isfinite(acc2.cpu().numpy()).all()
    assert (np.abs(out2[:, 0]) <= 1e-4).all() and out2[2, 9] == 1            # silog ~ 0 (x100 scale), one valid pixel in frame 2
    from bts_amd._lib import BtsHipError
    with pytest.raises(BtsHipError):
        E.gpu_compute_errors(pred.cpu(), gt.cpu(), "kitti", 1e-3, 80.0)
    with pytest.raises(BtsHipError):
        E.gpu_compute_errors(pred[:, :1], gt, "kitti", 1e-3, 80.0)          # size mismatch without kb-crop


@pytest.mark.gpu
def test_online_eval_loop_on_gpu_matches_host_protocol():
    """bts_main.py:193-275 end to end on a small model: the accumulator after the loop equals the per-sample host
    evaluation (prepare_eval + compute_errors on the model's own final_depth), averaged over the samples with valid depth."""
    from collections import namedtuple
    from bts_amd import bts as M, synth
    Params = namedtuple("Params", "encoder bts_size max_depth dataset")
    torch.manual_seed(3)
    model = M.BtsModel(Params("densenet121_bts", 512, 80.0, "kitti")).eval().cuda()
    args = SimpleNamespace(dataset="kitti", min_depth_eval=1e-3, max_depth_eval=80.0, do_kb_crop=False, garg_crop=True, eigen_crop=False)
    rng = np.random.default_rng(0)
    samples = []
    for i in range(4):
        img = torch.from_numpy(synth.image_batch(1, 64, 96, 50 + i))
        gt = torch.from_numpy(rng.uniform(0.0, 90.0, size=(1, 1, 64, 96)).astype(np.float32))
        samples.append(dict(image=img, focal=torch.from_numpy(synth.focal_values(1, "kitti", 50 + i)), depth=gt,
                            has_valid_depth=(i != 2)))
    got = E.online_eval(model, samples, args)
    want, cnt = np.zeros(9), 0
    with torch.no_grad():
        for smp in samples:
            if not smp["has_valid_depth"]:
                continue
            pred = model(smp["image"].cuda(), smp["focal"].cuda())[4].cpu().numpy().squeeze()
            gt = smp["depth"].numpy().squeeze()
            p2, valid = E.prepare_eval(pred, gt, "kitti", 1e-3, 80.0, garg_crop=True)
            want += np.asarray(E.compute_errors(gt[valid], p2[valid]))
            cnt += 1
    assert cnt == 3
    np.testing.assert_allclose(got[:9].numpy(), want / cnt, rtol=1e-10)
    assert got[9].item() == 1.0                    # the reference divides all ten entries by the count (bts_main.py:265)
