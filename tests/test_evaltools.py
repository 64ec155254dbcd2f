"""CPU: the output path / evaluation protocol helpers that mirror bts_test.py and bts_main.py."""
import os

import numpy as np
import pytest

from bts_amd import evaltools as E


def test_argfile_parsing(tmp_path):
    # same format as the reference's arguments_test_eigen.txt (one flag per line)
    f = tmp_path / "args.txt"
    f.write_text("--encoder densenet161_bts\n--data_path ../../dataset/kitti_dataset/\n--dataset kitti\n"
                 "--model_name bts_eigen_v2_pytorch_densenet161  # comment\n--input_height 352\n--input_width 1216\n"
                 "--max_depth 80\n--do_kb_crop\n\n")
    a = E.parse_args([str(f)])                       # len(argv)==1 -> arg file (bts_test.py:64-68)
    assert (a.encoder, a.dataset, a.input_height, a.input_width, a.max_depth, a.do_kb_crop) == \
        ("densenet161_bts", "kitti", 352, 1216, 80.0, True)
    assert a.model_name == "bts_eigen_v2_pytorch_densenet161" and a.bts_size == 512
    b = E.parse_args(["@" + str(f), "--garg_crop", "--max_depth_eval", "50"])
    assert b.garg_crop and b.max_depth_eval == 50 and b.do_kb_crop


def test_png16_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(1))
    d = rng.uniform(0.0, 80.0, size=(37, 53)).astype(np.float32)
    p = tmp_path / "d.png"
    E.write_depth_png16(str(p), d, "kitti")
    back = np.array(Image.open(p))
    assert back.dtype in (np.uint16, np.int32) and back.shape == d.shape
    assert np.array_equal(back.astype(np.uint16), (d * 256.0).astype(np.uint16))
    E.write_depth_png16(str(p), d / 10, "nyu")
    assert np.array_equal(np.array(Image.open(p)).astype(np.uint16), (d / 10 * 1000.0).astype(np.uint16))
    with pytest.raises(ValueError):
        E.write_depth_png16(str(p), d[None], "kitti")


def test_compute_errors_known_values():
    gt = np.array([1.0, 2.0, 4.0, 10.0])
    same = E.compute_errors(gt, gt)
    assert np.allclose(same[:6], 0) and same[6:] == [1.0, 1.0, 1.0]
    pred = gt * 2.0                                   # constant log offset: silog 0, thresholds 2 > 1.25^3
    m = dict(zip(E.EVAL_METRICS, E.compute_errors(gt, pred)))
    assert abs(m['silog']) < 1e-4 and m['d1'] == 0 and m['d2'] == 0 and m['d3'] == 0
    assert np.isclose(m['abs_rel'], 1.0) and np.isclose(m['log10'], np.log10(2)) and np.isclose(m['log_rms'], np.log(2))
    assert np.isclose(m['rms'], np.sqrt(np.mean(gt ** 2))) and np.isclose(m['sq_rel'], np.mean(gt))


def test_prepare_eval_masks_and_kb_crop():
    gt = np.full((375, 1242), 10.0, dtype=np.float32)
    gt[:100] = 0.0                                    # invalid (no lidar)
    pred = np.full((352, 1216), 5.0, dtype=np.float32)
    pred[0, 0], pred[0, 1], pred[0, 2] = np.inf, np.nan, 1e6
    p, valid = E.prepare_eval(pred, gt, "kitti", 1e-3, 80.0, do_kb_crop=True, garg_crop=True)
    assert p.shape == gt.shape
    top, left = 375 - 352, (1242 - 1216) // 2
    assert p[top, left] == 80.0 and p[top, left + 1] == np.float32(1e-3) and p[top, left + 2] == 80.0
    assert p[0, 0] == np.float32(1e-3)                # outside the crop: zero -> clamped to min
    assert not valid[:int(0.40810811 * 375)].any() and valid[200, 600]
    assert not valid[:, :int(0.03594771 * 1242)].any()
    _, v2 = E.prepare_eval(pred, gt, "kitti", 1e-3, 80.0, do_kb_crop=True, eigen_crop=True)
    assert v2[130, 600] and not valid[130, 600]       # the eigen crop starts higher up than garg's
    _, v3 = E.prepare_eval(np.ones((480, 640)), np.ones((480, 640)), "nyu", 1e-3, 10.0, eigen_crop=True)
    assert v3.sum() == (471 - 45) * (601 - 41)
