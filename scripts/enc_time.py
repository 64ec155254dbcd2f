import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from collections import namedtuple
from bts_amd import bts as M
P = namedtuple("Params", "encoder bts_size max_depth dataset")
mode = sys.argv[1]
if mode == "nocudnn":
    torch.backends.cudnn.enabled = False
enc = M.encoder(P("densenet161_bts", 512, 80.0, "kitti")).eval().cuda()
x = torch.randn(int(sys.argv[2]), 3, 352, 1216, device="cuda")
t0 = time.time()
with torch.no_grad():
    k = 0
    feats = [x]
    for name, v in enc.base_model._modules.items():
        if name.startswith("denseblock"):
            f = [feats[-1]]
            for ln, layer in v.items():
                f.append(layer(torch.cat(f, 1)))
                torch.cuda.synchronize()
                print("%s %s %.1fs" % (name, ln, time.time() - t0), flush=True)
            feats.append(torch.cat(f, 1))
        else:
            feats.append(v(feats[-1]))
        torch.cuda.synchronize()
        print("%s done %.1fs" % (name, time.time() - t0), flush=True)
    for i in range(3):
        torch.cuda.synchronize(); t1 = time.time()
        enc(x); torch.cuda.synchronize()
        print("steady pass %.1f ms" % (1e3 * (time.time() - t1)), flush=True)
