"""Per-shape workspace caches and packed-weight caches shared by a module and its replicas.

Two host-side mechanisms the fused forward depends on:

``WorkspaceCache`` -- the NHWC buffers of one (batch, height, width, device, sub-batch slot).  libbts_hip.so never
allocates, so a forward's launches carry raw pointers into these buffers, and a captured hipGraph keeps replaying
those pointers.  Eviction is therefore LRU over UNPINNED entries only (``max_entries`` of them are kept):
``graph.GraphedModel`` records which entries a capture touched (``recording()``) and pins them for as long as the graph
lives; pinned entries never count against the limit and are never handed back to the allocator under a live graph.

``PackCache`` -- kernel-layout copies of a module's weights (packed conv weights, folded BN vectors), rebuilt when
the parameters change (``load_state_dict``, an optimiser step, ``.cuda()``: detected through ``data_ptr``/``_version``).
``nn.DataParallel`` (the reference's inference protocol, bts_test.py:91) re-creates its replicas on every forward
with freshly broadcast parameter tensors, so a cache that lived on the replica and was keyed on the replica's own
tensors would re-pack every forward.  Here the cache object and a back-reference to the SOURCE module are created in
``__init__`` and reach every replica through the shallow ``__dict__`` copy ``replicate()`` makes; entries are keyed by
(device, fingerprint of the source module's tensors) and built from the replica's own tensors on its own device.
"""
from __future__ import annotations

import threading
from collections import OrderedDict
from contextlib import contextmanager
from typing import Callable, Dict, Hashable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

_tls = threading.local()


@contextmanager
def recording():
    """Collect the (cache, key) pairs every WorkspaceCache.get() touches inside the block (this thread only)."""
    prev = getattr(_tls, "touched", None)
    _tls.touched = touched = []
    try:
        yield touched
    finally:
        _tls.touched = prev


class WorkspaceCache:
    def __init__(self, max_entries: int = 8):
        self.max_entries = max_entries
        self._entries: "OrderedDict[Hashable, dict]" = OrderedDict()
        self._pins: Dict[Hashable, int] = {}
        self._lock = threading.Lock()

    def __reduce__(self):
        # copy.deepcopy(model) / torch.save(model): a copy starts with an EMPTY cache (buffers are scratch, and a lock
        # cannot be copied)
        return (WorkspaceCache, (self.max_entries,))

    def __len__(self):
        return len(self._entries)

    def __contains__(self, key):
        return key in self._entries

    def get(self, key: Hashable, factory: Callable[[], dict]) -> dict:
        touched = getattr(_tls, "touched", None)
        with self._lock:
            ws = self._entries.get(key)
            if ws is not None:
                self._entries.move_to_end(key)
        if ws is None:
            ws = factory()                                 # allocate outside the lock (device allocations can be slow)
            with self._lock:
                ws = self._entries.setdefault(key, ws)
                self._entries.move_to_end(key)
                self._evict(keep=key)
        if touched is not None:
            touched.append((self, key))
        return ws

    def _evict(self, keep):
        # the limit counts UNPINNED entries: pinned ones (live graphs) are extra, so a forward that needs S slots keeps
        # all S of them whatever the graphs hold
        free = [k for k in self._entries if self._pins.get(k, 0) == 0]      # oldest first
        excess = len(free) - self.max_entries
        for k in free:
            if excess <= 0:
                break
            if k != keep:
                del self._entries[k]
                excess -= 1

    def pin(self, key):
        with self._lock:
            self._pins[key] = self._pins.get(key, 0) + 1

    def unpin(self, key):
        with self._lock:
            n = self._pins.get(key, 0) - 1
            if n <= 0:
                self._pins.pop(key, None)
            else:
                self._pins[key] = n
            self._evict(keep=None)

    def pinned(self, key) -> bool:
        return self._pins.get(key, 0) > 0


def tensor_fingerprint(module: nn.Module) -> Tuple:
    """Cheap fingerprint of a module's parameters and buffers: changes when load_state_dict / an optimiser step /
    .cuda() / dist.broadcast_module touches them (in-place writes bump ``_version``; writes through ``.data`` do NOT,
    which is why this package never writes through ``.data`` -- callers that do must call ``invalidate_packs``).
    torch's FUSED optimisers (``AdamW(fused=True)``) do not bump ``_version`` either: a training forward of this
    package (train.begin_step) and every ``BtsModel.train()/eval()`` mode switch therefore call ``invalidate_packs``
    themselves, so a train -> eval hand-over never reuses packs of older values."""
    st = getattr(module, "_fp_state", None)
    if st is None:
        tensors = list(module.parameters()) + list(module.buffers())
    else:
        # a module that maintains ``_fp_state`` = [structure token, cached tensor list, token of the list] (BtsModel: the
        # token moves in _apply / load_state_dict(assign=True), the only paths that REPLACE tensor objects) spares the
        # recursive walk over ~1000 parameters and buffers that a replayed plan would otherwise pay per call (measured:
        # 4.6 -> 0.4 ms on the build container's CPU)
        if st[1] is None or st[2] != st[0]:
            st[1] = list(module.parameters()) + list(module.buffers())
            st[2] = st[0]
        tensors = st[1]
    return tuple((t.data_ptr(), t._version, t.device.index if t.device.type != "cpu" else -1) for t in tensors)


_generation = [0]


def invalidate_packs():
    """Force every PackCache to rebuild at its next use (for callers that rewrite weights through ``.data``)."""
    _generation[0] += 1


class PackCache:
    """See the module docstring.  ``owner``: the module whose __init__ creates the cache (the source module)."""

    def __init__(self, owner: nn.Module):
        self._origin = [owner]             # a list, so nn.Module.__setattr__ does not register it as a sub-module
        self._entries: Dict[str, Tuple] = {}

    def __reduce__(self):
        # deepcopy / pickle: the copy's cache is empty and its origin is the COPIED owner (the memo resolves the cycle)
        return (PackCache, (self._origin[0],))

    @property
    def origin(self) -> nn.Module:
        return self._origin[0]

    def get(self, device, build: Callable[[], object], key_modules: Optional[Callable[[nn.Module], Sequence[nn.Module]]] = None):
        origin = self.origin
        mods = [origin] if key_modules is None else list(key_modules(origin))
        key = (_generation[0],) + tuple(tensor_fingerprint(m) for m in mods)
        dev = str(device)
        hit = self._entries.get(dev)
        if hit is not None and hit[0] == key:
            return hit[1]
        pack = build()
        self._entries[dev] = (key, pack)
        return pack

    def clear(self):
        self._entries.clear()
