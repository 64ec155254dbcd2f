"""Shared helpers for decoder-level parity tests (GPU HIP path vs CPU oracle / goldens)."""
from collections import namedtuple

import numpy as np
import torch

from bts_amd import synth
from oracle import bts_oracle as O

Params = namedtuple("Params", "encoder bts_size max_depth dataset")
CONFIGS = {
    "K": ("densenet161_bts", 80.0, "kitti", 352, 1216),
    "N": ("resnext101_bts", 10.0, "nyu", 416, 544),
}
OUT_NAMES = ("depth_8x8_scaled", "depth_4x4_scaled", "depth_2x2_scaled", "reduc1x1", "final_depth", "iconv1")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_inputs(cname, B, H, W, seed):
    enc, md, ds, _, _ = CONFIGS[cname]
    feat = synth.ENCODER_CHANNELS[enc]
    feats = synth.encoder_features(feat, B, H, W, seed=seed)
    focal = synth.focal_values(B, ds, seed=seed)
    return [None] + [t(f) for f in feats[1:]], t(focal)


def oracle_run(cname, B, H, W, seed, state=None):
    enc, md, ds, _, _ = CONFIGS[cname]
    if state is None:
        state = O.state_from_numpy(synth.decoder_state(synth.ENCODER_CHANNELS[enc], 512, 0))
    feats, focal = make_inputs(cname, B, H, W, seed)
    with torch.no_grad():
        return O.decoder_forward(state, feats, focal, md, ds, want_intermediates=True)


def build_hip_decoder(cname, device="cuda"):
    """The product decoder (bts_amd.bts.bts) loaded with the PCG64(0) synthetic state."""
    from bts_amd import bts as M
    enc, md, ds, _, _ = CONFIGS[cname]
    feat = synth.ENCODER_CHANNELS[enc]
    dec = M.bts(Params(enc, 512, md, ds), feat, 512)
    sd = {k: (torch.tensor(v) if np.ndim(v) == 0 else t(v)) for k, v in synth.decoder_state(feat, 512, 0).items()}
    dec.load_state_dict(sd, strict=True)
    return dec.eval().to(device)


def hip_run(dec, cname, B, H, W, seed):
    feats, focal = make_inputs(cname, B, H, W, seed)
    feats = [None] + [f.cuda() for f in feats[1:]]
    with torch.no_grad():
        outs = dec(feats, focal.cuda())
    torch.cuda.synchronize()
    return outs


def singular_masks(inter, B, H, W):
    """Per LPG scale: True where |denominator| > 2e-3 (near-singular pixels are compared
    with an absolute tolerance or excluded; relative error is meaningless next to the +-1e-3 clamp)."""
    masks = {}
    for k, name in ((8, "plane_eq_8x8"), (4, "plane_eq_4x4"), (2, "plane_eq_2x2")):
        den = O.lpg_denominator(inter[name], k).unsqueeze(1).numpy()
        masks[k] = np.abs(den) > 2e-3
    return masks


def max_rel(a, b, mask=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))) if a.size else 0.0


def check_outputs(got, ref_outs, inter, rel_tol=1e-4, what=""):
    """north_star: <=1e-3 relative on the fp32 depth maps; asserted at the tighter 1e-4 guard.
    iconv1 crosses zero: rtol 1e-3 / atol 1e-4."""
    B, _, H, W = ref_outs[0].shape
    masks = singular_masks(inter, B, H, W)
    report = {}
    for (name, g, r, k) in ((OUT_NAMES[0], got[0], ref_outs[0], 8), (OUT_NAMES[1], got[1], ref_outs[1], 4),
                            (OUT_NAMES[2], got[2], ref_outs[2], 2)):
        g = g.detach().cpu().numpy()
        r = r.numpy()
        assert g.shape == r.shape, (name, g.shape, r.shape)
        mr = max_rel(g, r, masks[k])
        report[name] = mr
        assert mr <= rel_tol, "%s %s: max rel err %g (masked %d near-singular px)" % (what, name, mr, (~masks[k]).sum())
        # near-singular pixels: same sign/magnitude class (clamped plane => |depth| huge); allow abs tol on 1/depth
        ns = ~masks[k]
        if ns.any():
            fin = np.isfinite(r[ns]) & np.isfinite(g[ns])
            inv_err = np.abs(1.0 / g[ns][fin] - 1.0 / r[ns][fin])
            assert inv_err.size == 0 or inv_err.max() < 1e-2 * max(1.0, np.abs(1.0 / r[ns][fin]).max())
    for idx in (3, 4):
        g = got[idx].detach().cpu().numpy()
        r = ref_outs[idx].numpy()
        mr = max_rel(g, r)
        report[OUT_NAMES[idx]] = mr
        assert mr <= rel_tol, "%s %s: max rel err %g" % (what, OUT_NAMES[idx], mr)
    g = got[5].detach().cpu().numpy()
    r = ref_outs[5].numpy()
    err = np.abs(g - r)
    assert (err <= 1e-4 + 1e-3 * np.abs(r)).all(), "%s iconv1: max abs err %g" % (what, err.max())
    report["iconv1_max_abs"] = float(err.max())
    return report
