"""Output path and evaluation protocol of the reference scripts, for callers that switch to bts_amd
(SURVEY.md section 8(f)-4).  The metrics run as GPU reductions (``gpu_compute_errors`` / ``online_eval`` on
``bts_eval_depth_metrics_f32``, csrc/eval.hip); the NumPy functions below are the host-side statement of the same
protocol (arg files, PNG writer, and ``prepare_eval`` + ``compute_errors`` for callers that already hold NumPy maps).

  * ``make_arg_parser`` / ``parse_args``  -- the ``@argfile`` convention of bts_test.py:44-68, bts_eval.py:35-43
    (one ``--flag value`` per line, ``#`` comments), so ``arguments_test_eigen.txt``-style files load unchanged.
  * ``write_depth_png16``                -- depth -> uint16 PNG, x256 (KITTI) / x1000 (NYU), bts_test.py:203-215
    (cv2 is not available here: a minimal PNG encoder on zlib; compression level 0 as the reference).
  * ``prepare_eval`` + ``compute_errors`` -- kb-crop un-cropping, clamping, valid/garg/eigen masks and the nine
    metrics of bts_main.py:87-108, 221-251 (== bts_eval.py:81-102, 237-307).
  * ``predict``                           -- the timed inference loop of bts_test.py:127-147 on in-memory inputs.
  * ``eval_crop_rect`` / ``kb_crop_offsets`` -- the crop rectangles and the kb-crop paste offsets as integers.
  * ``gpu_compute_errors`` / ``online_eval`` -- the same metrics as ONE GPU pass per sample and the online-eval loop of
    bts_main.py:193-275 (accumulator of 9 sums + count, all-reduced over the ranks).
"""
from __future__ import annotations

import argparse
import struct
import time
import zlib
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

EVAL_METRICS = ['silog', 'abs_rel', 'log10', 'rms', 'sq_rel', 'log_rms', 'd1', 'd2', 'd3']   # bts_main.py:84


def _convert_arg_line_to_args(arg_line: str) -> Iterable[str]:
    """bts_eval.py:35-39, plus '#' comments as the training arg files use them."""
    line = arg_line.split('#', 1)[0]
    for arg in line.split():
        if arg.strip():
            yield arg


def make_arg_parser() -> argparse.ArgumentParser:
    """The inference/eval flags of bts_test.py:44-62 and bts_eval.py:45-66 (same names, types, defaults)."""
    p = argparse.ArgumentParser(description='BTS inference on bts_amd (MI355X).', fromfile_prefix_chars='@')
    p.convert_arg_line_to_args = _convert_arg_line_to_args
    p.add_argument('--model_name', type=str, default='bts_nyu_v2')
    p.add_argument('--encoder', type=str, default='densenet161_bts')
    p.add_argument('--data_path', type=str, default='')
    p.add_argument('--gt_path', type=str, default='')
    p.add_argument('--filenames_file', type=str, default='')
    p.add_argument('--input_height', type=int, default=480)
    p.add_argument('--input_width', type=int, default=640)
    p.add_argument('--max_depth', type=float, default=80)
    p.add_argument('--checkpoint_path', type=str, default='')
    p.add_argument('--dataset', type=str, default='nyu')
    p.add_argument('--do_kb_crop', action='store_true')
    p.add_argument('--save_lpg', action='store_true')
    p.add_argument('--bts_size', type=int, default=512)
    p.add_argument('--min_depth_eval', type=float, default=1e-3)
    p.add_argument('--max_depth_eval', type=float, default=80)
    p.add_argument('--eigen_crop', action='store_true')
    p.add_argument('--garg_crop', action='store_true')
    return p


def parse_args(argv: Sequence[str]):
    """``len(argv) == 1`` and not a flag  =>  treat it as an arg file (bts_test.py:64-68)."""
    p = make_arg_parser()
    if len(argv) == 1 and not argv[0].startswith('-'):
        argv = ['@' + argv[0].lstrip('@')]
    return p.parse_args(list(argv))


# ------------------------------------------------------------------------------------------- PNG
def _png_chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)


def depth_to_uint16(depth: np.ndarray, dataset: str) -> np.ndarray:
    """bts_test.py:203-208: metres -> 1/256 m (kitti family) or mm (others), truncated to uint16."""
    scale = 256.0 if dataset in ('kitti', 'kitti_benchmark', 'vkitti') else 1000.0
    return (np.asarray(depth, dtype=np.float32) * scale).astype(np.uint16)


def write_depth_png16(path: str, depth: np.ndarray, dataset: str) -> None:
    """16-bit grayscale PNG of ``depth_to_uint16(depth)`` (cv2.imwrite(..., PNG_COMPRESSION 0), bts_test.py:209)."""
    img = depth_to_uint16(depth, dataset)
    if img.ndim != 2:
        raise ValueError("write_depth_png16: expected a [H,W] depth map")
    h, w = img.shape
    rows = img.astype('>u2').tobytes()
    stride = 2 * w
    raw = b''.join(b'\x00' + rows[y * stride:(y + 1) * stride] for y in range(h))      # filter type 0 per scanline
    png = b'\x89PNG\r\n\x1a\n' + _png_chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 16, 0, 0, 0, 0)) \
        + _png_chunk(b'IDAT', zlib.compress(raw, 0)) + _png_chunk(b'IEND', b'')
    with open(path, 'wb') as f:
        f.write(png)


# ------------------------------------------------------------------------------------ evaluation
def compute_errors(gt: np.ndarray, pred: np.ndarray) -> List[float]:
    """bts_main.py:87-108: [silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3] over valid pixels."""
    gt = np.asarray(gt, dtype=np.float64)
    pred = np.asarray(pred, dtype=np.float64)
    thresh = np.maximum(gt / pred, pred / gt)
    d1, d2, d3 = [(thresh < 1.25 ** i).mean() for i in (1, 2, 3)]
    rms = np.sqrt(((gt - pred) ** 2).mean())
    log_rms = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    err = np.log(pred) - np.log(gt)
    silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    log10 = np.mean(np.abs(np.log10(pred) - np.log10(gt)))
    return [silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3]


def prepare_eval(pred_depth: np.ndarray, gt_depth: np.ndarray, dataset: str, min_depth_eval: float, max_depth_eval: float,
                 do_kb_crop: bool = False, garg_crop: bool = False, eigen_crop: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """bts_main.py:221-249: returns (pred, valid_mask) ready for ``compute_errors(gt[mask], pred[mask])``."""
    pred = np.array(pred_depth, dtype=np.float32, copy=True)
    gt = np.asarray(gt_depth)
    if do_kb_crop:
        height, width = gt.shape
        top, left = int(height - 352), int((width - 1216) / 2)
        full = np.zeros((height, width), dtype=np.float32)
        full[top:top + 352, left:left + 1216] = pred
        pred = full
    pred[pred < min_depth_eval] = min_depth_eval
    pred[pred > max_depth_eval] = max_depth_eval
    pred[np.isinf(pred)] = max_depth_eval
    pred[np.isnan(pred)] = min_depth_eval
    valid = np.logical_and(gt > min_depth_eval, gt < max_depth_eval)
    if garg_crop or eigen_crop:
        gh, gw = gt.shape
        m = np.zeros(valid.shape)
        if garg_crop:
            m[int(0.40810811 * gh):int(0.99189189 * gh), int(0.03594771 * gw):int(0.96405229 * gw)] = 1
        elif dataset == 'kitti':
            m[int(0.3324324 * gh):int(0.91351351 * gh), int(0.0359477 * gw):int(0.96405229 * gw)] = 1
        else:
            m[45:471, 41:601] = 1
        valid = np.logical_and(valid, m)
    return pred, valid


def predict(model, images, focals, batch: int = 1):
    """The inference loop of bts_test.py:127-147 on in-memory NCHW images: returns (five lists of [H,W] numpy depth
    maps in the order est, 8x8, 4x4, 2x2, 1x1, elapsed seconds)."""
    import torch
    pred_depths, p8, p4, p2, p1 = [], [], [], [], []
    t0 = time.time()
    with torch.no_grad():
        for i in range(0, len(images), batch):
            img = torch.as_tensor(np.stack(images[i:i + batch])).cuda()
            foc = torch.as_tensor(np.asarray(focals[i:i + batch], dtype=np.float32)).cuda()
            lpg8x8, lpg4x4, lpg2x2, reduc1x1, depth_est, _ = model(img, foc)
            for j in range(img.shape[0]):
                pred_depths.append(depth_est[j].cpu().numpy().squeeze())
                p8.append(lpg8x8[j].cpu().numpy().squeeze())
                p4.append(lpg4x4[j].cpu().numpy().squeeze())
                p2.append(lpg2x2[j].cpu().numpy().squeeze())
                p1.append(reduc1x1[j].cpu().numpy().squeeze())
    return pred_depths, p8, p4, p2, p1, time.time() - t0


# ------------------------------------------------------------------------- GPU reductions (csrc/eval.hip)
def kb_crop_offsets(gt_height: int, gt_width: int) -> Tuple[int, int]:
    """(top, left) at which the 352x1216 kb-cropped prediction sits in the ground-truth frame (bts_main.py:222-224)."""
    return int(gt_height - 352), int((gt_width - 1216) / 2)


def eval_crop_rect(gt_height: int, gt_width: int, dataset: str, garg_crop: bool, eigen_crop: bool) -> Tuple[int, int, int, int]:
    """(y0, y1, x0, x1) of the evaluation mask of bts_main.py:236-249 (the whole frame when no crop is requested)."""
    if garg_crop:
        return (int(0.40810811 * gt_height), int(0.99189189 * gt_height), int(0.03594771 * gt_width), int(0.96405229 * gt_width))
    if eigen_crop:
        if dataset == 'kitti':
            return (int(0.3324324 * gt_height), int(0.91351351 * gt_height), int(0.0359477 * gt_width), int(0.96405229 * gt_width))
        return (45, min(471, gt_height), 41, min(601, gt_width))
    return (0, gt_height, 0, gt_width)


def gpu_compute_errors(pred_depth, gt_depth, dataset: str, min_depth_eval: float, max_depth_eval: float,
                       do_kb_crop: bool = False, garg_crop: bool = False, eigen_crop: bool = False, accum=None):
    """prepare_eval + compute_errors for a BATCH of samples on the GPU: ``pred_depth`` [B,1,Hp,Wp] or [B,Hp,Wp] (the
    model's final_depth), ``gt_depth`` [B,1,Hg,Wg] or [B,Hg,Wg], both CUDA float32.  Returns a [B,10] float64 CUDA
    tensor: the nine measures in EVAL_METRICS order + the valid-pixel count per sample.  ``accum``: optional [10]
    float64 CUDA tensor, online_eval's running ``eval_measures`` (updated in place, frames in index order)."""
    import ctypes as C
    import torch
    from . import _lib, ops
    ops._need(pred_depth, "gpu_compute_errors")
    ops._need(gt_depth, "gpu_compute_errors")
    pred = pred_depth.reshape((pred_depth.shape[0],) + tuple(pred_depth.shape[-2:])).contiguous()
    gt = gt_depth.reshape((gt_depth.shape[0],) + tuple(gt_depth.shape[-2:])).contiguous()
    B, Hp, Wp = pred.shape
    if gt.shape[0] != B or gt.device != pred.device:
        raise _lib.BtsHipError("gpu_compute_errors: pred and gt must hold the same number of samples on one device")
    Hg, Wg = gt.shape[1:]
    top, left = kb_crop_offsets(Hg, Wg) if do_kb_crop else (0, 0)
    if not do_kb_crop and (Hp, Wp) != (Hg, Wg):
        raise _lib.BtsHipError("gpu_compute_errors: prediction %dx%d vs ground truth %dx%d (only do_kb_crop changes the size)"
                               % (Hp, Wp, Hg, Wg))
    if do_kb_crop and (top < 0 or left < 0 or top + Hp > Hg or left + Wp > Wg):
        raise _lib.BtsHipError("gpu_compute_errors: kb-cropped prediction %dx%d does not fit the %dx%d ground truth" % (Hp, Wp, Hg, Wg))
    y0, y1, x0, x1 = eval_crop_rect(Hg, Wg, dataset, garg_crop, eigen_crop)
    lib = _lib.load()
    ws = torch.empty((int(lib.bts_eval_ws_doubles(B, Hg, Wg)),), dtype=torch.float64, device=pred.device)
    out = torch.empty((B, 10), dtype=torch.float64, device=pred.device)
    if accum is not None and (accum.dtype != torch.float64 or accum.numel() != 10 or accum.device != pred.device
                              or not accum.is_contiguous()):
        raise _lib.BtsHipError("gpu_compute_errors: accum must be a contiguous [10] float64 tensor on the prediction's device")
    with torch.cuda.device(pred.device):
        rc = lib.bts_eval_depth_metrics_f32(ops._ptr(pred), B, Hp, Wp, ops._ptr(gt), Hg, Wg, top, left,
                                            float(min_depth_eval), float(max_depth_eval), y0, y1, x0, x1,
                                            ops._ptr(ws), ws.numel(), ops._ptr(out), ops._ptr(accum), ops._stream(pred))
    _lib.check(rc, "bts_eval_depth_metrics_f32")
    return out


def online_eval(model, samples, args, device="cuda"):
    """The online evaluation of bts_main.py:193-275 with the per-sample work on the GPU: for every sample
    ``dict(image [1,3,H,W], focal [1], depth [1,1,Hg,Wg], has_valid_depth)`` run the model, reduce the nine measures on
    the device (no map ever travels to the host), keep the 10-float accumulator (9 sums + count), all-reduce it over
    the ranks when torch.distributed is initialised (bts_main.py:258-260) and return the averaged measures as a CPU
    float64 tensor of 10 (the reference divides all ten entries by the count, bts_main.py:264-265).
    ``args``: needs dataset, min_depth_eval, max_depth_eval, do_kb_crop, garg_crop, eigen_crop."""
    import torch
    from . import dist as bdist
    acc = torch.zeros(10, dtype=torch.float64, device=device)
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for sample in samples:
            if not sample.get('has_valid_depth', True):
                continue                                        # bts_main.py:201-203
            image = sample['image'].to(device, non_blocking=True)
            focal = sample['focal'].to(device, non_blocking=True)
            gt = sample['depth'].to(device=device, dtype=torch.float32, non_blocking=True)
            pred = model(image, focal)[4]
            gpu_compute_errors(pred, gt, args.dataset, args.min_depth_eval, args.max_depth_eval, do_kb_crop=args.do_kb_crop,
                               garg_crop=args.garg_crop, eigen_crop=args.eigen_crop, accum=acc)
    if was_training:
        model.train()
    bdist.all_reduce_eval_measures(acc)
    cpu = acc.cpu()
    cnt = cpu[9].item()
    return cpu / cnt if cnt > 0 else cpu
