import sys, json, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from parity_util import *
from bts_amd import bts as M
import test_train_gpu as T
c = TRAIN_CASE
_, md, ds, _, _ = CONFIGS[c["cname"]]
g = np.load("tests/golden/decoder_train.npz")
dec = T._train_decoder()
feats, focal = make_inputs(c["cname"], c["B"], c["H"], c["W"], c["feat_seed"])
feats = [None] + [f.cuda().requires_grad_(True) for f in feats[1:]]
gt, mask = synth.train_targets(c["B"], c["H"], c["W"], md, c["target_seed"])
outs = dec(feats, focal.cuda())
loss = M.silog_loss(variance_focus=c["variance_focus"])(outs[4], t(gt).cuda(), t(mask).cuda())
loss.backward(); torch.cuda.synchronize()
noise, r64 = fp32_noise_floor()
rows = []
for n, p in dec.named_parameters():
    ref = r64["param_grads"][n].numpy(); scale = np.abs(ref).max()
    e = np.abs(p.grad.cpu().numpy() - ref).max() / scale
    flat = p.grad.cpu().numpy().reshape(-1)
    eg = np.abs(flat[g["gidx_"+n]] - g["gval_"+n]).max() / scale
    rows.append((n, float(e), float(noise[n]), float(eg), float(scale)))
rows.sort(key=lambda r: -r[1])
json.dump(rows, open("gpurun_out/train_noise.json", "w"), indent=0)
for r in rows[:15]: print("%-52s hip-vs-64 %.2e  cpu32-vs-64 %.2e  hip-vs-golden(samples) %.2e  scale %.1e" % r)
