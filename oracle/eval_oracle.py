"""TEST INFRASTRUCTURE ONLY (never imported by bts_amd/ or the timed region of bench.py).

CPU restatement, line for line and in the reference's own float32 NumPy arithmetic, of the per-sample evaluation of
minghanz/bts: ``compute_errors`` (pytorch/bts_main.py:87-108, identical in bts_eval.py:81-102 and
utils/eval_with_pngs.py:54-80) and the prediction clean-up + masks of ``online_eval`` (bts_main.py:221-251).

Parity status: **unpinned by reference outputs** -- the reference files that hold this code import cv2 / c3d at module
level (bts_main.py:48-54, bts_eval.py:27-30), which are absent here, so the functions cannot be imported to generate
fixtures, and the reference ships no evaluation fixtures of its own.  The restatement is pinned instead by
hand-derived values (tests/golden/eval_fixture.json: crop rectangles at the KITTI and NYU frame sizes worked out from
the constants of bts_main.py:236-249, and a 2x3 sample whose nine measures are computed by hand).
"""
import numpy as np


def compute_errors(gt, pred):
    """bts_main.py:87-108, verbatim arithmetic (inputs keep their dtype: float32 in the reference's callers)."""
    thresh = np.maximum((gt / pred), (pred / gt))
    d1 = (thresh < 1.25).mean()
    d2 = (thresh < 1.25 ** 2).mean()
    d3 = (thresh < 1.25 ** 3).mean()
    rms = (gt - pred) ** 2
    rms = np.sqrt(rms.mean())
    log_rms = (np.log(gt) - np.log(pred)) ** 2
    log_rms = np.sqrt(log_rms.mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    err = np.log(pred) - np.log(gt)
    silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    err = np.abs(np.log10(pred) - np.log10(gt))
    log10 = np.mean(err)
    return [silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3]


def eval_sample(pred_depth, gt_depth, dataset, min_depth_eval, max_depth_eval, do_kb_crop=False, garg_crop=False,
                eigen_crop=False):
    """bts_main.py:221-251 for one sample ([H,W] arrays): returns (measures, number of valid pixels)."""
    pred_depth = np.array(pred_depth, dtype=np.float32, copy=True)
    gt_depth = np.asarray(gt_depth)
    if do_kb_crop:                                                              # :221-227
        height, width = gt_depth.shape
        top_margin = int(height - 352)
        left_margin = int((width - 1216) / 2)
        pred_depth_uncropped = np.zeros((height, width), dtype=np.float32)
        pred_depth_uncropped[top_margin:top_margin + 352, left_margin:left_margin + 1216] = pred_depth
        pred_depth = pred_depth_uncropped
    pred_depth[pred_depth < min_depth_eval] = min_depth_eval                    # :229-232
    pred_depth[pred_depth > max_depth_eval] = max_depth_eval
    pred_depth[np.isinf(pred_depth)] = max_depth_eval
    pred_depth[np.isnan(pred_depth)] = min_depth_eval
    valid_mask = np.logical_and(gt_depth > min_depth_eval, gt_depth < max_depth_eval)   # :234
    if garg_crop or eigen_crop:                                                 # :236-249
        gt_height, gt_width = gt_depth.shape
        eval_mask = np.zeros(valid_mask.shape)
        if garg_crop:
            eval_mask[int(0.40810811 * gt_height):int(0.99189189 * gt_height),
                      int(0.03594771 * gt_width):int(0.96405229 * gt_width)] = 1
        elif eigen_crop:
            if dataset == 'kitti':
                eval_mask[int(0.3324324 * gt_height):int(0.91351351 * gt_height),
                          int(0.0359477 * gt_width):int(0.96405229 * gt_width)] = 1
            else:
                eval_mask[45:471, 41:601] = 1
        valid_mask = np.logical_and(valid_mask, eval_mask)
    return compute_errors(gt_depth[valid_mask], pred_depth[valid_mask]), int(valid_mask.sum())
