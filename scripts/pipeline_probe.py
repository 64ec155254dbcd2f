"""Probe: does overlapping consecutive batches (two model replicas, two streams, two graphs) beat back-to-back
graph replays?  `offset_ms` > 0 starts the second stream that much later, so that the two forwards stay out of phase
(one in its under-filled DenseNet block 3/4 layers while the other runs chip-filling decoder convolutions).
    python scripts/pipeline_probe.py [depth] [offset_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from bts_amd import synth, ops

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
offset_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda", 0)
params = B.Params("densenet161_bts", 512, 80.0, "kitti")
models = [B.build_model(params, dev, 0) for _ in range(depth)]
for m in models[1:]:
    m.load_state_dict(models[0].state_dict())
img = torch.from_numpy(synth.image_batch(16, 352, 1216, 1234)).to(dev)
foc = torch.from_numpy(synth.focal_values(16, "kitti", 1234)).to(dev)
streams = [torch.cuda.Stream(dev) for _ in range(depth)]
graphs, outs = [], []
with torch.no_grad():
    for m, st in zip(models, streams):
        with torch.cuda.stream(st):
            m(img, foc); m(img, foc)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            o = m(img, foc)
        graphs.append(g); outs.append(o)
    torch.cuda.synchronize()
    for K in (40, 40):
        t0 = time.perf_counter()
        for i in range(K):
            if i == 1 and offset_ms > 0:
                time.sleep(offset_ms * 1e-3)          # one-time phase offset between the streams
            with torch.cuda.stream(streams[i % depth]):
                graphs[i % depth].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("depth %d offset %.0f ms: %.3f ms/step  %.1f frames/s" % (depth, offset_ms, 1e3 * dt / K, 16 * K / dt), flush=True)
