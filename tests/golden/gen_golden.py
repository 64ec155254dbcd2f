#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference).  It imports
``/root/reference/pytorch/bts.py`` unmodified, replaces ``Tensor.cuda`` with an
identity (the reference hard-codes ``.cuda()`` inside LPG, bts.py:157,160), loads
PCG64-seeded synthetic parameters (``bts_amd.synth``) into the reference modules
in eval mode and records their outputs.  Inputs are NOT stored: tests regenerate
them from the same seeds.  Only data (arrays) is written; no reference source.

    python tests/golden/gen_golden.py
"""
import os
import sys
from collections import namedtuple

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/pytorch")

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # shim for bts.py:157,160
torch.set_num_threads(8)

import bts as ref  # noqa: E402  (the reference)
from bts_amd import synth  # noqa: E402

Params = namedtuple("Params", "encoder bts_size max_depth dataset")

CONFIGS = {
    # name: (encoder, max_depth, dataset, full H, full W)
    "K": ("densenet161_bts", 80.0, "kitti", 352, 1216),
    "N": ("resnext101_bts", 10.0, "nyu", 416, 544),
}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_state(module, state_np, prefix=""):
    sd = {}
    for k, v in state_np.items():
        if k.startswith(prefix):
            sd[k[len(prefix):]] = torch.tensor(v) if np.ndim(v) == 0 else t(v)
    module.load_state_dict(sd, strict=True)
    return module.eval()


# ------------------------------------------------------------------ (1) LPG tables
def lpg_cases():
    out = {}
    for k in (2, 4, 8):
        # hand-picked: one row of cells, n1=n2=0 so den == n3 everywhere in the cell
        dens = np.array([5e-4, -5e-4, 0.0, 2e-3, -2e-3, 1e-3, -1e-3, 0.5, 1.0], dtype=np.float32)
        pe = np.zeros((1, 4, 1, dens.size), dtype=np.float32)
        pe[0, 2, 0, :] = dens
        pe[0, 3, 0, :] = 0.5
        out["hand_%d_in" % k] = pe
        # orientation probes: den = n1*u + n3 (varies along columns), den = n2*v + n3 (rows)
        po = np.zeros((2, 4, 2, 3), dtype=np.float32)
        po[0, 0] = 1.0
        po[0, 2] = 10.0
        po[0, 3] = 1.0
        po[1, 1] = 1.0
        po[1, 2] = 10.0
        po[1, 3] = 1.0
        out["orient_%d_in" % k] = po
        # random unit normals with n3 >= 0.5 (as the reduction epilogue produces) + distances
        rng = np.random.Generator(np.random.PCG64(100 + k))
        B, h, w = 2, 5, 7
        theta = rng.uniform(0, np.pi / 3, size=(B, h, w))
        phi = rng.uniform(0, 2 * np.pi, size=(B, h, w))
        pr = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta),
                       rng.uniform(0.1, 80.0, size=(B, h, w))], axis=1).astype(np.float32)
        out["rand_%d_in" % k] = pr
        m = ref.local_planar_guidance(k)
        for name in ("hand", "orient", "rand"):
            x = t(out["%s_%d_in" % (name, k)])
            with torch.no_grad():
                y = m(x, torch.ones(x.shape[0]))
            out["%s_%d_out" % (name, k)] = y.numpy()
            out["%s_%d_absmin" % (name, k)] = np.float32(m.abs_min.item())
    return out


# ---------------------------------------------------------- (2) per-module goldens
MOD_SHAPES = [(2, 11, 19), (1, 13, 17)]  # (B,h,w); d=24 exceeds both maps


def module_cases():
    out = {}
    for cname, (enc, max_depth, dataset, _, _) in CONFIGS.items():
        feat = synth.ENCODER_CHANNELS[enc]
        state = synth.decoder_state(feat, 512, seed=0)
        nf = 512
        reducs = {
            "reduc8x8": (nf // 4, nf // 4, False),
            "reduc4x4": (nf // 4, nf // 8, False),
            "reduc2x2": (nf // 8, nf // 16, False),
            "reduc1x1": (nf // 16, nf // 32, True),
        }
        for name, (cin, cout, fin) in reducs.items():
            for md in (80.0, 10.0):
                m = load_state(ref.reduction_1x1(cin, cout, md, is_final=fin), state, name + ".")
                for si, (B, h, w) in enumerate(MOD_SHAPES):
                    rng = np.random.Generator(np.random.PCG64(2000 + si))
                    x = rng.standard_normal(size=(B, cin, h, w), dtype=np.float32)
                    with torch.no_grad():
                        y = m(t(x))
                    out["%s_%s_md%d_s%d" % (cname, name, int(md), si)] = y.numpy()
        aspp = {
            "daspp_3": (nf // 2, 3, False),
            "daspp_6": (nf // 2 + nf // 4 + feat[2], 6, True),
            "daspp_12": (nf + feat[2], 12, True),
            "daspp_18": (nf + nf // 4 + feat[2], 18, True),
            "daspp_24": (nf + nf // 2 + feat[2], 24, True),
        }
        for name, (cin, dil, fbn) in aspp.items():
            m = load_state(ref.atrous_conv(cin, nf // 4, dil, apply_bn_first=fbn), state, name + ".")
            for si, (B, h, w) in enumerate(MOD_SHAPES):
                rng = np.random.Generator(np.random.PCG64(3000 + si))
                x = rng.standard_normal(size=(B, cin, h, w), dtype=np.float32)
                with torch.no_grad():
                    y = m(t(x))
                out["%s_%s_s%d" % (cname, name, si)] = y.numpy()
    return out


# ------------------------------------------------------- (3) decoder end-to-end
OUT_NAMES = ("depth_8x8_scaled", "depth_4x4_scaled", "depth_2x2_scaled", "reduc1x1", "final_depth", "iconv1")


def run_decoder(cname, B, H, W, feat_seed, want_den=False):
    enc, max_depth, dataset, _, _ = CONFIGS[cname]
    feat = synth.ENCODER_CHANNELS[enc]
    state = synth.decoder_state(feat, 512, seed=0)
    p = Params(enc, 512, max_depth, dataset)
    dec = load_state(ref.bts(p, feat, 512), state)
    feats = synth.encoder_features(feat, B, H, W, seed=feat_seed)
    focal = synth.focal_values(B, dataset, seed=feat_seed)
    planes = {}
    hooks = []
    if want_den:      # the plane equations the reference's own LPG layers receive (bts.py:254, 268, 281)
        for k, m in ((8, dec.lpg8x8), (4, dec.lpg4x4), (2, dec.lpg2x2)):
            hooks.append(m.register_forward_pre_hook(lambda mod, args, k=k: planes.__setitem__(k, args[0].detach().clone())))
    with torch.no_grad():
        outs = dec([None] + [t(f) for f in feats[1:]], t(focal))
    for h in hooks:
        h.remove()
    absmin = [dec.lpg8x8.abs_min.item(), dec.lpg4x4.abs_min.item(), dec.lpg2x2.abs_min.item()]
    if want_den:
        from oracle import bts_oracle as O
        dens = {k: O.lpg_denominator(planes[k], k).unsqueeze(1).numpy() for k in planes}
        return [o.numpy() for o in outs], np.asarray(absmin, dtype=np.float32), dens
    return [o.numpy() for o in outs], np.asarray(absmin, dtype=np.float32)


def decoder_small():
    out = {}
    for cname in CONFIGS:
        outs, absmin = run_decoder(cname, 2, 64, 96, feat_seed=4321)
        for n, o in zip(OUT_NAMES, outs):
            out["%s_%s" % (cname, n)] = o
        out["%s_abs_min" % cname] = absmin
    return out


N_SAMPLES = 4096


def decoder_full():
    out = {}
    for cname, (_, _, _, H, W) in CONFIGS.items():
        outs, absmin, dens = run_decoder(cname, 1, H, W, feat_seed=1234, want_den=True)
        rng = np.random.Generator(np.random.PCG64(999))
        for n, o in zip(OUT_NAMES, outs):
            flat = o.reshape(-1)
            idx = np.sort(rng.choice(flat.size, size=N_SAMPLES, replace=False)).astype(np.int64)
            out["%s_%s_idx" % (cname, n)] = idx
            out["%s_%s_val" % (cname, n)] = flat[idx]
            fin = flat[np.isfinite(flat)]
            out["%s_%s_stats" % (cname, n)] = np.asarray(
                [fin.min(), fin.max(), fin.astype(np.float64).mean(), np.abs(fin).astype(np.float64).mean(),
                 float(flat.size - fin.size)], dtype=np.float64)
        # |LPG denominator| of the REFERENCE's own plane equations at the sampled pixels of the three LPG outputs: the
        # test masks the near-singular samples (|den| <= 2e-3) exactly instead of allowing a fraction to fail
        for k, n in ((8, OUT_NAMES[0]), (4, OUT_NAMES[1]), (2, OUT_NAMES[2])):
            out["%s_%s_absden" % (cname, n)] = np.abs(dens[k].reshape(-1)[out["%s_%s_idx" % (cname, n)]]).astype(np.float32)
        out["%s_abs_min" % cname] = absmin
    return out


# ------------------------------------------------- (4) training step (batch-stat BN + backward)
TRAIN_GRAD_SAMPLES = 64


def decoder_train():
    """Reference decoder in train() mode: forward with batch-statistic BN, silog loss on final_depth
    (bts_main.py:476-481 protocol, variance_focus 0.85), backward.  Records the loss, the gradient w.r.t. the
    encoder taps (full), per-parameter gradient norms + seeded samples, and the BN running stats after the step."""
    out = {}
    cname, B, H, W = "K", 2, 64, 96
    enc, max_depth, dataset, _, _ = CONFIGS[cname]
    feat = synth.ENCODER_CHANNELS[enc]
    state = synth.decoder_state(feat, 512, seed=0)
    dec = load_state(ref.bts(Params(enc, 512, max_depth, dataset), feat, 512), state).train()
    feats = [t(f).requires_grad_(True) for f in synth.encoder_features(feat, B, H, W, seed=4321)[1:]]
    focal = t(synth.focal_values(B, dataset, seed=4321))
    gt, mask = synth.train_targets(B, H, W, max_depth, seed=77)
    outs = dec([None] + feats, focal)
    loss = ref.silog_loss(variance_focus=0.85)(outs[4], t(gt), t(mask))
    loss.backward()
    out["loss"] = np.asarray(loss.item(), dtype=np.float64)
    for n, o in zip(OUT_NAMES, outs):
        out["out_" + n] = o.detach().numpy()
    for i, f in enumerate(feats):
        out["grad_feat%d" % (i + 1)] = f.grad.numpy()
    rng = np.random.Generator(np.random.PCG64(555))
    for name, prm in dec.named_parameters():
        g = prm.grad.detach().numpy().reshape(-1)
        idx = np.sort(rng.choice(g.size, size=min(TRAIN_GRAD_SAMPLES, g.size), replace=False)).astype(np.int64)
        out["gidx_" + name] = idx
        out["gval_" + name] = g[idx]
        out["gnorm_" + name] = np.asarray(np.sqrt((g.astype(np.float64) ** 2).sum()))
    for name, buf in dec.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            out["buf_" + name] = buf.detach().numpy()
    out["abs_min"] = np.asarray([dec.lpg8x8.abs_min.item(), dec.lpg4x4.abs_min.item(), dec.lpg2x2.abs_min.item()],
                                dtype=np.float32)
    return out


def main():
    if "--only-full" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "decoder_full_samples.npz"), **decoder_full())
        print("decoder_full_samples.npz", os.path.getsize(os.path.join(HERE, "decoder_full_samples.npz")) // 1024, "KiB")
        return
    if "--only-train" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "decoder_train.npz"), **decoder_train())
        print("decoder_train.npz", os.path.getsize(os.path.join(HERE, "decoder_train.npz")) // 1024, "KiB")
        return
    np.savez_compressed(os.path.join(HERE, "decoder_train.npz"), **decoder_train())
    np.savez_compressed(os.path.join(HERE, "lpg_tables.npz"), **lpg_cases())
    np.savez_compressed(os.path.join(HERE, "modules_small.npz"), **module_cases())
    np.savez_compressed(os.path.join(HERE, "decoder_small.npz"), **decoder_small())
    np.savez_compressed(os.path.join(HERE, "decoder_full_samples.npz"), **decoder_full())
    meta = "torch %s; reference /root/reference/pytorch/bts.py; seeds: params PCG64(0), see gen_golden.py\n" % torch.__version__
    with open(os.path.join(HERE, "GOLDEN_META.txt"), "w") as f:
        f.write(meta)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
