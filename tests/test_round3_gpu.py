"""Round-3 GPU tests: the launch declaration (fill_frames, precision) as a per-model attribute instead of a process
global, and the kernels / host paths added this round."""
import threading

import numpy as np
import pytest
import torch

from bts_amd import synth
from parity_util import Params, t

pytestmark = pytest.mark.gpu


def _model(enc="densenet161_bts", dataset="kitti", seed=3):
    from bts_amd import bts as M
    torch.manual_seed(seed)
    m = M.BtsModel(Params(enc, 512, 80.0 if dataset == "kitti" else 10.0, dataset))
    sd = {k: (torch.tensor(v) if np.ndim(v) == 0 else t(v))
          for k, v in synth.decoder_state(synth.ENCODER_CHANNELS[enc], 512, 0).items()}
    m.decoder.load_state_dict(sd, strict=True)
    return m.eval()


def _aspp_kernels(m, img, foc):
    from bts_amd import ops
    tr = ops.KernelTrace()
    ops.set_trace(tr)
    try:
        outs = [o.clone() for o in m(img, foc)]
    finally:
        ops.set_trace(None)
    return outs, sorted({r[0] for r in tr.records if r[1].startswith("aspp")})


def test_two_models_keep_their_own_launch_declaration():
    """Two models with different ``fill_frames`` in ONE process, called alternately and from two threads at once: each
    keeps the bits of its own setting (the declaration is a per-call thread-local scope, not a process global), the
    two settings really differ, and a model left at the default (None) follows the batch of the call: B = 1 gets the
    single-frame latency setting (the ASPP convolutions split K), B = 4 the library default (they do not)."""
    a = _model().cuda()
    b = _model().cuda()
    b.load_state_dict(a.state_dict())
    a.sub_batches = b.sub_batches = 1
    a.fill_frames, b.fill_frames = 8, 2
    H, W = 352, 1216
    img = t(synth.image_batch(1, H, W, 31)).cuda()
    foc = t(synth.focal_values(1, "kitti", 31)).cuda()
    with torch.no_grad():
        ra, ka = _aspp_kernels(a, img, foc)
        rb, kb = _aspp_kernels(b, img, foc)
        assert not any("splitk" in k for k in ka) and any("splitk" in k for k in kb), (ka, kb)
        assert not all(torch.equal(x, y) for x, y in zip(ra, rb)), "the two declarations give the same bits"
        for x, y in zip(ra, rb):
            assert (x - y).abs().max().item() / y.abs().max().item() < 2e-5
        for _ in range(2):                                             # alternating calls: no leakage either way
            assert all(torch.equal(x, y) for x, y in zip(a(img, foc), ra))
            assert all(torch.equal(x, y) for x, y in zip(b(img, foc), rb))
        # two threads, two streams, at the same time
        bad = []

        def work(model, ref):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st), torch.no_grad():
                    for _ in range(3):
                        outs = model(img, foc)
                        st.synchronize()
                        if not all(torch.equal(x, y) for x, y in zip(outs, ref)):
                            bad.append("bits changed under a concurrent model with another declaration")
            except Exception as e:                                      # noqa: BLE001
                bad.append(repr(e))

        th = [threading.Thread(target=work, args=(a, ra)), threading.Thread(target=work, args=(b, rb))]
        [x.start() for x in th]
        [x.join() for x in th]
        assert not bad, bad
        # default (None): by the batch of the call
        c = _model().cuda()
        c.load_state_dict(a.state_dict())
        c.sub_batches = 1
        assert c.fill_frames is None
        r1, k1 = _aspp_kernels(c, img, foc)
        assert all(torch.equal(x, y) for x, y in zip(r1, rb)), "B=1 at the default must be the fill_frames=2 path"
        img4 = t(synth.image_batch(4, H, W, 31)).cuda()
        foc4 = t(synth.focal_values(4, "kitti", 31)).cuda()
        r4, k4 = _aspp_kernels(c, img4, foc4)
        assert not any("splitk" in k for k in k4), k4
        a4 = a(img4, foc4)
        assert all(torch.equal(x, y) for x, y in zip(r4, a4)), "B=4 at the default must be the fill_frames=8 path"


def test_plans_and_graphs_follow_the_launch_declaration():
    """A recorded plan and a captured graph bake the declaration into every descriptor: after ``conv_precision`` or
    ``fill_frames`` changes on the model, neither may replay the old recording (set_conv_precision did not
    age plans / graphs)."""
    from bts_amd.graph import GraphedModel
    m = _model("densenet121_bts").cuda()
    m.sub_batches = 1
    img = t(synth.image_batch(1, 96, 128, 5)).cuda()
    foc = t(synth.focal_values(1, "kitti", 5)).cuda()
    with torch.no_grad():
        ref = {}
        for prec in ("fp32", "bf16x3"):
            m.conv_precision = prec
            ref[prec] = [o.clone() for o in m(img, foc)]
        assert not all(torch.equal(x, y) for x, y in zip(ref["fp32"], ref["bf16x3"]))
        gm = GraphedModel(m)
        m.use_plans = True
        for prec in ("fp32", "bf16x3", "fp32"):
            m.conv_precision = prec
            for _ in range(2):                                  # record / capture, then replay
                assert all(torch.equal(x, y) for x, y in zip(m(img, foc), ref[prec])), "plan replayed another precision"
            m.use_plans = False
            for _ in range(2):
                assert all(torch.equal(x, y) for x, y in zip(gm(img, foc), ref[prec])), "graph replayed another precision"
            m.use_plans = True
        assert gm.captures == 2 and len(gm._graphs) == 2         # one graph per declaration, the fp32 one reused
        m.use_plans = False
        m.conv_precision = "fp32"
        m.fill_frames = 8
        r8 = [o.clone() for o in m(img, foc)]
        g8 = gm(img, foc)
        assert all(torch.equal(x, y) for x, y in zip(g8, r8)) and gm.captures == 3
