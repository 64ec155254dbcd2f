"""hipGraph replay of a whole BtsModel forward.

Nothing in libbts_hip.so allocates or synchronises and every workspace is preallocated per shape, so one
``model(image, focal)`` is a fixed DAG of ~460 kernel launches on up to four streams.  Capturing it once and replaying
it removes the host launch path (eager B=16: 54.3 ms/step, graph replay: 47.6 ms on MI355X).  This is the product's
answer to "a tracing compiler": HIP streams + one graph per input shape.

    gm = GraphedModel(model)            # model: BtsModel in eval mode on a GPU
    outs = gm(image, focal)             # first call per shape: two eager warm-ups + capture; then replay

CONTRACT
  * the returned tensors are the graph's static outputs -- they are overwritten by the next call with the same
    shape.  Consume (or clone) them before calling again; ``bts_test.py`` does exactly that (``.cpu().numpy()`` per
    frame).
  * a graph replays raw pointers: to the NHWC workspaces of its shape and to the PACKED copies of the weights made at
    capture time.  Both are kept alive and consistent here: the workspaces a capture touched are pinned in their
    ``WorkspaceCache`` until the graph is dropped (so a third input shape can never hand them back to the allocator
    under a live graph), and every call compares a fingerprint of the model's parameters and buffers
    (``data_ptr`` / ``_version``) with the one taken at capture -- after ``load_state_dict``, an optimiser step or a
    ``dist.broadcast_module`` the graph is re-captured instead of replaying stale weights (the online-eval pattern of
    bts_main.py:193-260: train steps, ``eval()``, forward).  Writes through ``.data`` bump no version: call
    ``bts_amd.workspace.invalidate_packs()`` after them.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Tuple

import torch

from . import workspace


class _Entry:
    __slots__ = ("graph", "image", "focal", "outs", "fingerprint", "pins", "packs")


class GraphedModel(torch.nn.Module):
    def __init__(self, model: torch.nn.Module, max_shapes: int = 4):
        super().__init__()
        self.model = model
        self.max_shapes = max_shapes
        self._graphs: "OrderedDict[Tuple, _Entry]" = OrderedDict()
        self.captures = 0                     # how many captures happened (tests / diagnostics)

    def _fingerprint(self):
        return (workspace._generation[0],) + workspace.tensor_fingerprint(self.model)

    def _drop(self, key):
        e = self._graphs.pop(key, None)
        if e is not None:
            for cache, k in e.pins:
                cache.unpin(k)

    def _capture(self, image, focal) -> _Entry:
        e = _Entry()
        e.image = image.clone()
        e.focal = focal.to(device=image.device, dtype=torch.float32).clone() if isinstance(focal, torch.Tensor) else focal
        dev = image.device
        with torch.no_grad():
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):                 # weight packing, workspaces, lazy module init happen here
                    self.model(e.image, e.focal)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            e.fingerprint = self._fingerprint()
            g = torch.cuda.CUDAGraph()
            with workspace.recording() as touched:
                with torch.cuda.graph(g):
                    e.outs = self.model(e.image, e.focal)
        e.graph = g
        e.pins = list(dict.fromkeys(touched))       # unique (cache, key) pairs, order kept
        for cache, k in e.pins:
            cache.pin(k)
        self.captures += 1
        return e

    def forward(self, image: torch.Tensor, focal):
        if self.model.training or not image.is_cuda:
            return self.model(image, focal)
        fshape = tuple(focal.shape) if isinstance(focal, torch.Tensor) else None
        # a capture bakes in the model's launch declaration (fill_frames by the batch unless pinned, precision)
        key = (tuple(image.shape), str(image.device), str(image.dtype), fshape,
               getattr(self.model, "fill_frames", None), getattr(self.model, "conv_precision", 0))
        entry = self._graphs.get(key)
        if entry is not None and entry.fingerprint != self._fingerprint():
            self._drop(key)                         # weights changed since capture: the graph points at stale packs
            entry = None
        if entry is None:
            while len(self._graphs) >= self.max_shapes:
                self._drop(next(iter(self._graphs)))        # least recently used graph, with its workspace pins
            entry = self._capture(image, focal)
            self._graphs[key] = entry
        self._graphs.move_to_end(key)
        entry.image.copy_(image)
        if isinstance(focal, torch.Tensor):
            entry.focal.copy_(focal.to(device=entry.focal.device, dtype=torch.float32))
        entry.graph.replay()
        return entry.outs
