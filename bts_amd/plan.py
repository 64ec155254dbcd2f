"""Recorded forwards ("plans"): one crossing of the ctypes boundary per forward instead of one per launch.

``BtsModel.use_plans = True`` makes the fused eval forward record, once per (input shape, device, sub-batch slot), the
exact sequence of libbts_hip.so calls it issues -- ~120 launches for one 352x1216 frame -- as an array of ``bts_op``
(include/bts_hip.h) and replay it afterwards with a single ``bts_plan_run`` call.  The arguments of a forward depend
only on the shape: workspaces and packed weights are fixed, only the caller's tensors (image, focal, the six outputs,
the three abs_min scalars) move, and those pointer fields are found by address range after the recording and patched
per call.  Unlike a hipGraph (bts_amd/graph.py) a plan imposes no static-output contract: every call returns fresh
tensors, as the eager forward does.  The reference's inference loop is batch 1, eager (pytorch/bts_test.py:127-147):
that is the regime where the ~460 us of Python/ctypes per-launch overhead per frame matter.

A plan is dropped (and re-recorded on the next call) when the model's parameters change (fingerprint of
``data_ptr``/``_version``, as for packed weights) -- and it pins its workspaces in their WorkspaceCache while it lives.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, workspace
from . import ops as ops_mod
from ._lib import BtsHipError, ConvDesc

# entry point -> (BTS_OP_* kind, member name in the bts_op union); argument order = the C prototypes (stream excluded)
_KINDS = {
    "bts_conv_fwd_f32": (1, "conv"),
    "bts_reduc_fwd_f32": (2, "reduc"),
    "bts_reduc_lpg_fwd_f32": (3, "reduc_lpg"),
    "bts_lpg_fused_fwd_f32": (4, "lpg_fused"),
    "bts_nchw_to_nhwc_f32": (5, "nchw_to_nhwc"),
    "bts_nhwc_to_nchw_f32": (6, "nhwc_to_nchw"),
    "bts_maxpool3x3s2_nhwc_f32": (7, "maxpool"),
    "bts_bn_relu_avgpool2_nhwc_f32": (8, "avgpool"),
    "bts_get_depth_f32": (9, "get_depth"),
    "bts_lpg_fwd_f32": (10, "lpg"),
    "bts_upconv_combine_f32": (11, "upconv_combine"),
}

_structs: Dict[str, type] = {}
_BtsOp = None
_BtsPatch = None


def _build_types():
    """ctypes mirrors of bts_op / bts_plan_patch, derived from the argtypes _lib.load() declares (so the field order is
    the prototypes' by construction; the C side spells the same lists out as structs)."""
    global _BtsOp, _BtsPatch
    if _BtsOp is not None:
        return
    lib = _lib.load_real()
    members = []
    for fname, (kind, member) in _KINDS.items():
        if member == "conv":
            cls = ConvDesc
        else:
            argtypes = getattr(lib, fname).argtypes[:-1]
            cls = type("Args_" + member, (C.Structure,), {"_fields_": [("a%d" % k, t) for k, t in enumerate(argtypes)]})
        _structs[fname] = cls
        members.append((member, cls))

    class U(C.Union):
        _fields_ = members

    class BtsOp(C.Structure):
        _fields_ = [("kind", C.c_int), ("failed_code", C.c_int), ("u", U)]

    class BtsPatch(C.Structure):
        _fields_ = [("op", C.c_int), ("field_offset", C.c_int), ("slot", C.c_int), ("reserved", C.c_int), ("delta", C.c_long)]

    _BtsOp, _BtsPatch = BtsOp, BtsPatch
    lib.bts_plan_run.restype = C.c_int
    lib.bts_plan_run.argtypes = [C.POINTER(BtsOp), C.c_int, C.POINTER(BtsPatch), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p]


def _pointer_fields(cls) -> List[Tuple[int, int]]:
    """[(byte offset inside the struct, count)] of its pointer-typed fields."""
    out = []
    for name, t in cls._fields_:
        off = getattr(cls, name).offset
        if t is C.c_void_p:
            out.append((off, 1))
        elif isinstance(t, type) and issubclass(t, C.Array) and t._type_ is C.c_void_p:
            out.append((off, t._length_))
    return out


class _Recorder:
    def __init__(self):
        self.calls = []          # (entry point name, filled struct, stream handle)
        self.foreign = []        # entry points that were called but cannot be part of a plan


class _Proxy:
    """Stands in for the loaded library while a recording is active: forwards every call, keeps a copy of its arguments."""

    def __init__(self, lib, rec: _Recorder):
        _build_types()
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        real = getattr(self._lib, name)
        if not name.startswith("bts_") or name in ("bts_hip_error_string", "bts_hip_abi_version", "bts_conv_plan_f32",
                                                  "bts_eval_ws_doubles", "bts_bn_train_ws_floats", "bts_pack_weights_blocks"):
            return real
        if name not in _KINDS:
            def passthrough(*args):
                self._rec.foreign.append(name)
                return real(*args)
            return passthrough
        cls = _structs[name]

        def recorded(*args):
            stream = args[-1]
            stream = stream.value if isinstance(stream, C.c_void_p) else stream
            if cls is ConvDesc:
                st = ConvDesc.from_buffer_copy(bytes(args[0]._obj))
            else:
                st = cls()
                for k, v in enumerate(args[:-1]):
                    setattr(st, "a%d" % k, v.value if isinstance(v, C.c_void_p) else v)
            rc = real(*args)
            self._rec.calls.append((name, st, stream or 0))
            return rc
        return recorded


class Plan:
    def __init__(self, calls, io_tensors: Sequence[torch.Tensor], stream: int):
        _build_types()
        n = len(calls)
        self.ops = (_BtsOp * n)()
        patches = []
        ranges = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in io_tensors]
        u_off = _BtsOp.u.offset
        for i, (name, st, strm) in enumerate(calls):
            if strm != stream:
                raise BtsHipError("plan: the recorded forward used more than one stream")
            kind, member = _KINDS[name]
            self.ops[i].kind = kind
            C.memmove(C.byref(self.ops[i], u_off), C.byref(st), C.sizeof(st))
            for off, count in _pointer_fields(type(st)):
                for j in range(count):
                    ptr = C.c_void_p.from_buffer(st, off + 8 * j).value
                    if not ptr:
                        continue
                    for slot, (lo, hi) in enumerate(ranges):
                        if lo <= ptr < hi:
                            patches.append((i, u_off + off + 8 * j, slot, ptr - lo))
                            break
        self.n_ops = n
        self.patches = (_BtsPatch * max(len(patches), 1))()
        for k, (i, off, slot, delta) in enumerate(patches):
            self.patches[k].op, self.patches[k].field_offset, self.patches[k].slot, self.patches[k].delta = i, off, slot, delta
        self.n_patches = len(patches)
        self.n_slots = len(io_tensors)
        self.slots = (C.c_void_p * self.n_slots)()
        self._lock = threading.Lock()

    def run(self, io_tensors: Sequence[torch.Tensor], stream: int):
        # bts_plan_run patches the caller's pointers into the shared `ops` array in place: two threads replaying the same
        # plan (same model, device, slot and stream) must not interleave between patching and enqueueing
        with self._lock:
            for k, t in enumerate(io_tensors):
                self.slots[k] = t.data_ptr()
            rc = _lib.load_real().bts_plan_run(self.ops, self.n_ops, self.patches, self.n_patches, self.slots, self.n_slots,
                                               C.c_void_p(stream))
            if rc != 0:
                bad = next((i for i in range(self.n_ops) if self.ops[i].failed_code != 0), -1)
                _lib.check(rc, "bts_plan_run (op %d of %d)" % (bad, self.n_ops))


class PlanCache:
    """Plans of one BtsModel, keyed by (input shape, device, sub-batch slot); shared with DataParallel replicas."""

    def __init__(self, max_plans: int = 32):
        self.max_plans = max_plans
        self._plans: Dict[tuple, tuple] = {}
        self._lock = threading.Lock()
        self.recordings = 0

    def __reduce__(self):
        return (PlanCache, (self.max_plans,))

    def _drop(self, key):
        e = self._plans.pop(key, None)
        if e is not None:
            for cache, k in e[2]:
                cache.unpin(k)

    def clear(self):
        for k in list(self._plans):
            self._drop(k)

    def forward(self, model, x: torch.Tensor, focal, slot: int, outs: Optional[Sequence[torch.Tensor]]):
        """model._forward_native through a plan.  x: [b,3,H,W] CUDA tensor; focal: [b] tensor or None; outs: the six
        preallocated result tensors (batch slices of full-batch outputs) or None."""
        dev = x.device
        x = x.float().contiguous()
        has_focal = isinstance(focal, torch.Tensor)
        if has_focal:
            focal = focal.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        b, _, H, W = x.shape
        nf16 = model.decoder.num_features // 16
        if outs is None:
            outs = [torch.empty((b, c, H, W), dtype=torch.float32, device=dev) for c in (1, 1, 1, 1, 1, nf16)]
        am = torch.empty(3, dtype=torch.float32, device=dev)
        io = [x] + ([focal] if has_focal else []) + list(outs) + [am]
        stream = torch.cuda.current_stream(dev).cuda_stream
        origin = model._origin[0]
        fp = (workspace._generation[0], workspace.tensor_fingerprint(origin))
        # the launch declaration in force (fill_frames, precision) is baked into every recorded descriptor
        key = (tuple(x.shape), str(dev), slot, has_focal, stream, ops_mod.current_launch_config())
        entry = self._plans.get(key)
        if entry is not None and entry[1] != fp:
            with self._lock:
                self._drop(key)
            entry = None
        dec = model.decoder
        if entry is None:
            # two eager passes first (weight packing, workspaces), then the recorded one
            model._forward_native_eager(x, focal, slot, outs)
            rec = _Recorder()
            with workspace.recording() as touched:
                with _lib.recording(_Proxy(_lib.load_real(), rec)):
                    model._forward_native_eager(x, focal, slot, outs)
            if rec.foreign:
                raise BtsHipError("plan: the forward called %s, which a plan cannot replay" % sorted(set(rec.foreign)))
            rec_am = [dec.lpg8x8.abs_min, dec.lpg4x4.abs_min, dec.lpg2x2.abs_min]
            # the recorded pass wrote its abs_min scalars into three fresh 0-d tensors: treat them as three more slots
            plan = Plan(rec.calls, io[:-1] + rec_am, stream)
            pins = list(dict.fromkeys(touched))
            for cache, k in pins:
                cache.pin(k)
            with self._lock:
                while len(self._plans) >= self.max_plans:
                    self._drop(next(iter(self._plans)))
                self._plans[key] = (plan, fp, pins)
                self.recordings += 1
            return tuple(outs)
        plan = entry[0]
        plan.run(io[:-1] + [am[0:1], am[1:2], am[2:3]], stream)
        dec.lpg8x8.abs_min, dec.lpg4x4.abs_min, dec.lpg2x2.abs_min = am[0], am[1], am[2]
        return tuple(outs)
