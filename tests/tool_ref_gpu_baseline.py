"""Tooling (not a test): time the reference formulation of the decoder -- the oracle's plain torch ops, i.e. what
pytorch/bts.py launches -- on the GPU through PyTorch-ROCm (MIOpen), next to the HIP decoder, B=16 352x1216.
MIOpen has no gfx950 find-db in this image, so the first pass JIT-compiles every conv config (minutes).
    python tests/tool_ref_gpu_baseline.py [B]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch

from bts_amd import synth
from oracle import bts_oracle as O
from parity_util import build_hip_decoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 352, 1216
feat = synth.ENCODER_CHANNELS["densenet161_bts"]
state = {k: v.cuda() for k, v in O.state_from_numpy(synth.decoder_state(feat, 512, 0)).items()}
fe = synth.encoder_features(feat, B, H, W, 1234)
feats = [None] + [torch.from_numpy(f).cuda() for f in fe[1:]]
focal = torch.from_numpy(synth.focal_values(B, "kitti", 1234)).cuda()
t0 = time.time()


def ref():
    return O.decoder_forward(state, feats, focal, 80.0, "kitti")


with torch.no_grad():
    for i in range(3):
        ref()
        torch.cuda.synchronize()
        print("[%.0fs] reference-on-GPU pass %d done" % (time.time() - t0, i), flush=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ref()
    e.record(); torch.cuda.synchronize()
    t_ref = s.elapsed_time(e) / 5
    dec = build_hip_decoder("K")
    for _ in range(2):
        dec(feats, focal)
    s.record()
    for _ in range(5):
        dec(feats, focal)
    e.record(); torch.cuda.synchronize()
    t_hip = s.elapsed_time(e) / 5
print("decoder B=%d %dx%d: reference torch ops on PyTorch-ROCm/MIOpen %.2f ms | bts_amd HIP decoder %.2f ms | ratio %.2fx"
      % (B, H, W, t_ref, t_hip, t_ref / t_hip), flush=True)
