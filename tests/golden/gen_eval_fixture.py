#!/usr/bin/env python3
"""Writes tests/golden/eval_fixture.json: HAND-DERIVED expectations for the evaluation path.
Nothing here imports the reference or the code under test: the crop rectangles are the products of the constants in
pytorch/bts_main.py:236-249 with the frame sizes, truncated by int() as the reference does, written out as literals;
the nine measures of the small sample are closed forms of its four valid (gt, pred) pairs."""
import json
import math
import os

ln2, ln125 = math.log(2.0), math.log(1.25)
fixture = {
    "_about": "hand-derived; see gen_eval_fixture.py for the derivations",
    "crop_rects": [
        # [gt_height, gt_width, dataset, garg_crop, eigen_crop, y0, y1, x0, x1]
        # 0.40810811*375=153.04 0.99189189*375=371.96 0.03594771*1242=44.65 0.96405229*1242=1197.35
        [375, 1242, "kitti", True, False, 153, 371, 44, 1197],
        # 0.3324324*375=124.66 0.91351351*375=342.57 0.0359477*1242=44.65
        [375, 1242, "kitti", False, True, 124, 342, 44, 1197],
        # 0.40810811*370=151.0000007 0.99189189*370=366.9999993 0.03594771*1226=44.07 0.96405229*1226=1181.93
        [370, 1226, "kitti", True, False, 151, 366, 44, 1181],
        # 0.3324324*370=122.99999 -> 122 ; 0.91351351*370=337.9999987 -> 337 ; 0.0359477*1226=44.07
        [370, 1226, "kitti", False, True, 122, 337, 44, 1181],
        [480, 640, "nyu", False, True, 45, 471, 41, 601],          # bts_main.py:247, constants
        [480, 640, "nyu", False, False, 0, 480, 0, 640],
    ],
    "kb_crop_offsets": [[375, 1242, 23, 13], [370, 1226, 18, 5], [376, 1241, 24, 12]],   # [H, W, H-352, int((W-1216)/2)]
    "sample": {
        # gt 0 and 100 fall outside (1e-3, 80): four valid pairs (2,2) (4,2) (8,16) (1,1.25)
        "gt": [[2.0, 4.0, 8.0], [1.0, 0.0, 100.0]],
        "pred": [[2.0, 2.0, 16.0], [1.25, 5.0, 5.0]],
        "min_depth_eval": 1e-3, "max_depth_eval": 80.0, "valid": 4,
        "measures": {
            "silog": 100.0 * math.sqrt((2 * ln2 ** 2 + ln125 ** 2) / 4 - (ln125 / 4) ** 2),   # err = 0, -ln2, ln2, ln1.25
            "abs_rel": (0 + 0.5 + 1.0 + 0.25) / 4,
            "log10": (2 * math.log10(2.0) + math.log10(1.25)) / 4,
            "rms": math.sqrt((0 + 4 + 64 + 0.0625) / 4),                                       # = 4.125
            "sq_rel": (0 + 1 + 8 + 0.0625) / 4,
            "log_rms": math.sqrt((2 * ln2 ** 2 + ln125 ** 2) / 4),
            "d1": 0.25,      # thresh = 1, 2, 2, 1.25: only 1 < 1.25
            "d2": 0.5,       # 1 and 1.25 < 1.5625
            "d3": 0.5,       # 2 is not < 1.953125
        },
        # clean-up of the prediction (bts_main.py:229-232): NaN -> min, +inf -> max, -inf -> min, 0 -> min, 200 -> max
        "cleanup_in": ["nan", "inf", "-inf", 0.0, 200.0, 5.0],
        "cleanup_out": [1e-3, 80.0, 1e-3, 1e-3, 80.0, 5.0],
    },
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_fixture.json")
json.dump(fixture, open(path, "w"), indent=1)
print("wrote", path)
