#!/usr/bin/env python3
"""Headline benchmark: depth frames/s of BtsModel.forward -- the DenseNet161 encoder AND the decoder hot path both on
the hand-written HIP kernels of libbts_hip.so -- at B=16 per GPU, 3x352x1216 fp32 synthetic KITTI-shape input
(BASELINE.json configs[1]); one process per GPU, RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                    # starts the N ranks itself (the reference: mp.spawn, bts_main.py:843-847)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W       # or under torchrun: RANK / LOCAL_RANK / WORLD_SIZE from the env
    python bench.py --gpus 8 --global-batch 64      # BASELINE.json configs[3]: B=64 sharded 8 x 8 (strong scaling)

Weak scaling by default (--batch frames per GPU); --global-batch G fixes the total and shards it.  `n_gpus` in the
JSON line is the number of ranks RCCL actually formed; a mismatch with --gpus is an error, never a silent 1-GPU run.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant hand-written
kernel, HIP-event timed on its launch stream; `roofline.survey_8d` = SURVEY.md section 8(d)'s own definitions for the
ASPP branches and the fused reduction + LPG launches; `roofline.groups` / `.kernels` = every call site / kernel family)
and `cpu_baseline` (the CPU oracle on the host cores).  `config.fill_frames` is the frames-per-launch declaration the
library's tile / split-K choices were sized for (the per-GPU batch, at most 16).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from collections import namedtuple

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # only read by --encoder-backend miopen (A/B leg; MIOpen is off otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

Params = namedtuple("Params", "encoder bts_size max_depth dataset")

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32-input peak
PEAK_HBM_GBS = 8000.0


_T0 = time.time()


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def build_model(params, device, seed=0):
    """Random-init encoder (kaiming) + PCG64(seed) synthetic decoder state (bts_amd.synth)."""
    from bts_amd import bts as M, synth
    torch.manual_seed(seed)
    model = M.BtsModel(params)
    feat = synth.ENCODER_CHANNELS[params.encoder]
    sd = {k: (torch.tensor(v) if np.ndim(v) == 0 else torch.from_numpy(v.copy()))
          for k, v in synth.decoder_state(feat, params.bts_size, seed).items()}
    model.decoder.load_state_dict(sd, strict=True)
    # keep random-init encoder activations O(1): eval-BN with unit stats does not renormalise 160 layers
    return model.eval().to(device)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json:
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc runs of this same command).  Counters cannot be
    collected from inside the benchmark process itself, so the file carries the hash of the kernel sources it was
    measured on (`_meta.csrc_sha16`, scripts/pmc_traffic.py); a file measured on OTHER kernel sources yields
    traffic = null (stale) instead of a wrong number."""
    import glob
    import re
    from bts_amd import _lib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        tab = json.load(open(files[-1]))
    except Exception:
        return None
    meta = tab.get("_meta", {})
    src = {"source": os.path.basename(files[-1]), "profile_csrc_sha16": meta.get("csrc_sha16"),
           "profile_commit": meta.get("commit"), "current_csrc_sha16": _lib.source_hash()}
    if meta.get("csrc_sha16") != src["current_csrc_sha16"]:
        src["hbm_bytes_per_launch"] = None
        src["stale"] = True
        return src
    want = trace_to_rocprof_name(kernel)
    for k, v in tab.items():
        if k != "_meta" and want is not None and re.fullmatch(want, k):
            src["hbm_bytes_per_launch"] = int(v["hbm_MB_per_launch"] * 1e6)
            src["rocprof_kernel"] = k
            return src
    return None


def trace_to_rocprof_name(kernel):
    """Regex for the rocprofv3 (demangled, spaces removed) name of a kernel label used by ops.KernelTrace:
    conv_fwd_kernel<BM,BN,nhwc|nchw[,splitk]> -> conv_fwd_kernel<BM,BN,WM,WN,MF,false|true,0>;
    conv_halo_kernel<BN,kK,nhwc|nchw[,tail]> -> conv_halo_kernel<BN,WM,WN,MF,K,false|true,false|true>."""
    import re
    m = re.match(r"conv_fwd_kernel<(\d+),(\d+),(nhwc|nchw)(,splitk)?>", kernel)
    if m:
        return r"conv_fwd_kernel<%s,%s,\d+,\d+,\d+,%s(,0)?>" % (m.group(1), m.group(2), "true" if m.group(3) == "nchw" else "false")
    m = re.match(r"conv_stem_kernel<(\d+)>", kernel)
    if m:
        return r"conv_stem_kernel<%s>" % m.group(1)
    m = re.match(r"conv_(wino|halo_emu)_kernel<([\d,k]+)>", kernel)
    if m:
        return r"conv_%s_kernel<%s>" % (m.group(1), m.group(2).replace("k", ""))
    m = re.match(r"conv1x1_kernel<(\d+),(\d+)>", kernel)
    if m:
        return r"conv1x1_kernel<%s,%s(,true|,false)?>" % (m.group(1), m.group(2))      # 3rd parameter: single weight buffer
    m = re.match(r"conv_halo_kernel<(\d+),k(\d),(nhwc|nchw)(,tail)?(,w8)?(,dil)?>", kernel)
    if m:
        wm = "8" if m.group(5) else (r"\d+" if m.group(1) != "48" else "4")      # the 48-wide tile has a 4- and an 8-wave variant
        # 8th parameter: single weight buffer; 9th: dilation of the tile (1, or 3 / 6 / 12 for the dilated ASPP tiles)
        return r"conv_halo_kernel<%s,%s,\d+,\d+,%s,%s,%s(,true|,false)?%s>" % (
            m.group(1), wm, m.group(2), "true" if m.group(3) == "nchw" else "false", "true" if m.group(4) else "false",
            r",(3|6|12)" if m.group(6) else r"(,1)?")
    return None


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box
    exposes all host cores to os.cpu_count() but grants a 1-GPU job a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(params, H, W, seconds_budget=15.0, gpu_frame0=None, extra_frame0=None):
    """The oracle (CPU restatement, oracle/bts_oracle.py) + the same torch encoder on the host cores.
    Bounded sample: B=1 frames of the same 352x1216 workload until ~seconds_budget is spent."""
    from bts_amd import bts as M, synth
    from oracle import bts_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads (os.cpu_count()=%s)" % (cores, os.cpu_count()))
    torch.manual_seed(0)
    enc = M.encoder(params).eval()
    state = O.state_from_numpy(synth.decoder_state(synth.ENCODER_CHANNELS[params.encoder], params.bts_size, 0))
    img = torch.from_numpy(synth.image_batch(1, H, W, 1234))
    focal = torch.from_numpy(synth.focal_values(1, params.dataset, 1234))
    times = []
    parity = None
    extra_parity = {}

    def compare(gpu_set, outs, inter):
        names = ("depth_8x8_scaled", "depth_4x4_scaled", "depth_2x2_scaled", "reduc1x1", "final_depth")
        par = {"tolerance": 1e-3, "frame": 0}
        near = 0
        for j, nm in enumerate(names):
            g = gpu_set[j].double().cpu().numpy()
            r = outs[j][0:1].double().numpy()
            mask = np.ones(r.shape, dtype=bool)
            if j < 3:
                k = (8, 4, 2)[j]
                den = O.lpg_denominator(inter["plane_eq_%dx%d" % (k, k)], k).unsqueeze(1).numpy()
                mask = np.abs(den) > 2e-3          # near the +-1e-3 clamp relative error is meaningless
                near += int((~mask).sum())
            par[nm] = float(np.max(np.abs(g - r)[mask] / np.maximum(np.abs(r)[mask], 1e-30)))
        ic = np.abs(gpu_set[5].double().cpu().numpy() - outs[5][0:1].double().numpy())
        par["iconv1_max_abs"] = float(ic.max())
        par["near_singular_lpg_px_masked"] = near
        par["ok"] = bool(max(par[n] for n in names) <= 1e-3)
        return par, names

    with torch.no_grad():
        t_all = time.perf_counter()
        for i in range(40):
            t0 = time.perf_counter()
            feats = enc(img)
            ref = O.decoder_forward(state, feats, focal, params.max_depth, params.dataset, want_intermediates=(i == 0))
            dt = time.perf_counter() - t0
            if i == 0 and gpu_frame0 is not None:
                # parity gate of THIS run: frame 0 of the GPU batch (same PCG64 image, same weights) vs the CPU oracle
                outs, inter = ref
                parity, names = compare(gpu_frame0, outs, inter)
                log("parity vs CPU oracle (frame 0): " + ", ".join("%s %.2e" % (n, parity[n]) for n in names))
                for tag, gset in (extra_frame0 or {}).items():
                    extra_parity[tag], _ = compare(gset, outs, inter)
                    log("parity vs CPU oracle (frame 0, %s): " % tag + ", ".join("%s %.2e" % (n, extra_parity[tag][n]) for n in names))
            log("cpu frame %d: %.2f s" % (i, dt))
            if i > 0:
                times.append(dt)
            if time.perf_counter() - t_all > seconds_budget and len(times) >= 2:
                break
    med = float(np.median(times))
    out = {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "B=1 x %d timed frames (1 warm-up) of the same 3x%dx%d fp32 workload, torch %s CPU encoder + "
                     "oracle decoder, median" % (len(times), H, W, torch.__version__)}
    return out, parity, extra_parity


def baseline_config_label(enc, b, G, world, scaling, H, W):
    """Which BASELINE.json configuration a run corresponds to (the metric is quoted on configs[1])."""
    if (enc, b, H, W, scaling) == ("densenet161_bts", 16, 352, 1216, "weak"):
        return " (BASELINE.json configs[1]%s)" % ("" if world == 1 else ", weak-scaled to %d GPUs" % world)
    if (enc, G, H, W, scaling) == ("densenet161_bts", 64, 352, 1216, "strong"):
        return " (BASELINE.json configs[3]: B=64 batch-sharded%s)" % (
            " over 8 GPUs" if world == 8 else "; run here on %d GPU%s of the 8 it names" % (world, "" if world == 1 else "s"))
    if (enc, b, H, W) == ("resnext101_bts", 16, 416, 544):
        return " (BASELINE.json configs[2], not the headline configuration)"
    return " (not a BASELINE.json configuration)"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="number of ranks (one process per GPU).  Outside torchrun, N > 1 makes this process start the N "
                         "ranks itself; under torchrun it must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: total frames, sharded contiguously over the ranks (64 over 8 GPUs = BASELINE.json "
                         "configs[3]); must be a multiple of --gpus.  0 = weak scaling with --batch per GPU")
    ap.add_argument("--height", type=int, default=352)
    ap.add_argument("--width", type=int, default=1216)
    ap.add_argument("--encoder", default="densenet161_bts")
    ap.add_argument("--dataset", choices=["kitti", "nyu"], default="kitti",
                    help="decoder head: kitti (max_depth 80, focal scaling) or nyu (max_depth 10), bts.py:289-291")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-emulated-leg", action="store_true",
                    help="skip the secondary bf16x3-emulated-fp32 measurement (N=1 only; never the headline value)")
    ap.add_argument("--no-gather", action="store_true", help="skip the per-step all-gather of the 5 depth maps (N>1)")
    ap.add_argument("--streams", type=int, default=4,
                    help="run the per-GPU batch as this many concurrent sub-batches on separate HIP streams (one graph): "
                         "frames are independent, so under-filled launches of one sub-batch overlap the other's")
    ap.add_argument("--fill-frames", type=int, default=0,
                    help="pin BtsModel.fill_frames (frames per launch the library's split-K / tile choices are sized for, "
                         "bts_conv_desc.fill_frames).  0 = leave the model's default: by the batch of the call "
                         "(B <= 2 -> 2, B <= 11 -> 8, else 16)")
    ap.add_argument("--decoder-only", action="store_true", help="time only the decoder hot path on encoder-shaped features")
    ap.add_argument("--encoder-backend", choices=["hip", "aten", "miopen"], default="hip",
                    help="hip: DenseNet encoder on the HIP conv kernel (default); aten: torch encoder on ATen's native "
                         "conv path; miopen: torch encoder on MIOpen (no gfx950 find-db in this image: the first pass "
                         "JIT-compiles ~160 conv configs for >7 min)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port for the self-started ranks (0 = pick a free one)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="exercise ONLY the multi-rank plumbing on CPU (gloo): launcher, rendezvous, shard plan, barrier + "
                         "max-over-ranks timing, the packed all-gather.  No hot-path compute, no frames/s value")
    return ap


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this very script, one per GPU (the reference
    starts its workers the same way: mp.spawn(main_worker, nprocs=ngpus), bts_main.py:843-847, and rendezvous over
    tcp://127.0.0.1, bts_main.py:295).  The parent never initialises the GPU (torch.cuda.device_count() does not, on
    this image); it waits for the ranks, relays rank 0's JSON line and exits with the first failing rank's code."""
    n = args.gpus
    if not args.launcher_selftest:
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d requested but only %d GPU%s visible on this node; refusing to run fewer ranks than "
                  "asked for (the JSON line's n_gpus must be the number of ranks that really ran)" % (n, have, "" if have == 1 else "s"),
                  file=sys.stderr)
            return 2
    port = args.master_port or _free_port()
    procs = []
    # rank 0's stdout (the one JSON line) goes to a temporary file, read after the ranks have exited: the parent never
    # blocks on a pipe, so the 50 ms poll below is what notices a dying rank -- whichever rank it is -- and stops the
    # others (a rank > 0 that dies would otherwise leave rank 0 inside a collective until the RCCL timeout)
    import tempfile
    out0_file = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0_file if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = set(range(n))
        deadline = None                                    # set once a rank has failed: survivors get 10 s to obey SIGTERM
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
                        for q in pending:
                            procs[q].terminate()
                        deadline = time.monotonic() + 10.0
            if deadline is not None and pending and time.monotonic() > deadline:
                for q in pending:
                    procs[q].kill()
                deadline = time.monotonic() + 3600.0
            if pending:
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    if rc == 0:
        out0_file.seek(0)
        sys.stdout.write(out0_file.read().decode())
        sys.stdout.flush()
    out0_file.close()
    return rc


def shard_plan(args, world, rank=0):
    """(frames of THIS rank, frames of the largest rank, global batch, scaling label).  Weak: --batch per GPU.  Strong:
    --global-batch split into contiguous blocks, the first G % world ranks one frame longer (DataParallel.scatter
    order, bts_test.py:91; bts_amd.dist.shard_range) -- a batch the ranks do not divide is sharded unevenly, the gather
    pads and trims it (dist.DepthGather)."""
    if args.global_batch > 0:
        from bts_amd import dist as bdist
        if args.global_batch < world:
            raise SystemExit("bench.py: --global-batch %d is smaller than the %d ranks" % (args.global_batch, world))
        lo, hi = bdist.shard_range(args.global_batch, rank, world)
        return hi - lo, -(-args.global_batch // world), args.global_batch, "strong"
    return args.batch, args.batch, args.batch * world, "weak"


def launcher_selftest(args, world, rank):
    """CPU / gloo stand-in for the N-rank run: everything around the hot path that bench.py does for N > 1."""
    from bts_amd import dist as bdist
    dist.init_process_group(backend="gloo")
    formed = dist.get_world_size()
    if os.environ.get("BTS_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        # failure injection (tests/test_bench_launcher.py): this rank dies after the rendezvous while the others go on
        # into a collective they can never finish -- the launcher must notice and stop them
        os._exit(7)
    b, b_max, G, scaling = shard_plan(args, formed, rank)
    lo, hi = bdist.shard_range(G, rank, formed)
    assert hi - lo == b <= b_max, "contiguous shards, at most one frame apart"
    m = torch.nn.Linear(4, 3)
    with torch.no_grad():
        m.weight.fill_(float(rank + 1))
    bdist.broadcast_module(m, src=0)
    assert float(m.weight.detach()[0, 0]) == 1.0, "weights come from rank 0"
    outs = [torch.full((b, 1, 2, 3), float(10 * rank + i)) for i in range(6)]
    dist.barrier()
    t0 = time.perf_counter()
    gathered, work = bdist.all_gather_depths(outs, 5, async_op=True, global_batch=G)
    work.wait()
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0 + 0.001 * rank], dtype=torch.float64)
    mine = float(tt.item())
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    maps = bdist.unshard_depths(gathered, G)
    starts = [bdist.shard_range(G, r, formed) for r in range(formed)]
    ok = all(tuple(maps[i].shape) == (G, 1, 2, 3) for i in range(5)) and \
        all(float(maps[i][f, 0, 0, 0]) == 10 * r + i for i in range(5) for r, (a, e) in enumerate(starts) for f in range(a, e))
    line = {"selftest": "launcher", "n_gpus": formed, "requested_gpus": args.gpus, "scaling": scaling,
            "global_batch": G, "batch_per_gpu": b, "shard": [lo, hi], "gather_ok": bool(ok),
            "elapsed_is_max_over_ranks": bool(float(tt.item()) >= mine), "value": None, "backend": "gloo"}
    dist.barrier()
    dist.destroy_process_group()
    return line


def main():
    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL does at init) are
    # sent to stderr for the whole run; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    args = build_parser().parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torchrun: this process only starts the ranks (it must not touch the GPU) and relays rank 0's line
        os.dup2(real_stdout, 1)
        sys.exit(launch_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher environment says WORLD_SIZE=%d; refusing to report a run whose "
              "n_gpus differs from what was asked for" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.launcher_selftest:
        line = launcher_selftest(args, world, rank)
        if rank == 0:
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        return
    use_dist = world > 1 or bool(os.environ.get("BTS_BENCH_FORCE_DIST"))   # FORCE: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if torch.cuda.device_count() <= local_rank:
            print("bench.py: rank %d has no GPU %d on this node (%d visible)" % (rank, local_rank, torch.cuda.device_count()),
                  file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            print("bench.py: RCCL formed %d ranks, --gpus asked for %d" % (dist.get_world_size(), args.gpus), file=sys.stderr)
            sys.exit(2)
        world = dist.get_world_size()          # n_gpus in the JSON line = the ranks RCCL actually formed
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.enabled = args.encoder_backend == "miopen"

    from bts_amd import dist as bdist, ops, synth
    is_kitti = args.dataset == "kitti"
    params = Params(args.encoder, 512, 80.0 if is_kitti else 10.0, "kitti" if is_kitti else "nyu")
    B, B_max, G, scaling = shard_plan(args, world, rank)
    H, W = args.height, args.width
    log("building model %s" % args.encoder)
    S = max(1, args.streams)
    while S > 1 and B % S:
        S -= 1
    # The model declares how many frames share the chip (BtsModel.fill_frames -> bts_conv_desc.fill_frames).  Left at
    # None it follows the batch of the call in three classes (ops.auto_fill_frames: B >= 12 -> 16, the B frames that
    # bench.py keeps in flight as S concurrent sub-batches); --fill-frames / $BTS_CONV_FILL_FRAMES (A/B runs) pin it.
    pinned_fill = int(os.environ["BTS_CONV_FILL_FRAMES"]) if "BTS_CONV_FILL_FRAMES" in os.environ else (args.fill_frames or None)
    fill_frames = pinned_fill if pinned_fill else ops.auto_fill_frames(B)
    model = build_model(params, device, seed=0)
    model.native_encoder = args.encoder_backend == "hip"
    model.sub_batches = S
    model.fill_frames = model.decoder.fill_frames = pinned_fill
    bdist.broadcast_module(model, src=0)            # RCCL broadcast of ~188 MB, once
    log("model on %s (%d concurrent sub-batch%s)" % (device, S, "es" if S > 1 else ""))

    image = torch.from_numpy(synth.image_batch(B, H, W, 1234 + rank)).to(device)
    focal = torch.from_numpy(synth.focal_values(B, params.dataset, 1234 + rank)).to(device)
    feats_static = None
    if args.decoder_only:
        fe = synth.encoder_features(synth.ENCODER_CHANNELS[args.encoder], B, H, W, 1234 + rank)
        feats_static = [None] + [torch.from_numpy(f).to(device) for f in fe[1:]]

    # N > 1: the five depth maps of every step are all-gathered (one collective).  The model writes them straight into
    # the persistent send buffer (BtsModel.output_buffers = views of dist.DepthGather's packed buffer: no pack copy, no
    # per-step allocation); two slots -- two captured graphs -- alternate, so step i+1 computes into the other slot
    # while the collective of step i is still reading its own.
    gather = use_dist and not args.no_gather and not args.decoder_only
    dg = bdist.DepthGather(5, B_max, H, W, device, slots=2) if gather else None
    n_slots = 2 if gather else 1

    def forward(slot=0):
        if feats_static is not None:
            return model.decoder(feats_static, focal)
        model.output_buffers = dg.outputs(slot, B) if dg is not None else None
        return model(image, focal)

    use_graph = not args.no_graph
    graph, outs = None, None
    graphs, outs_slot = [None] * n_slots, [None] * n_slots
    with torch.no_grad():
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(2):                     # eager passes: MIOpen kernel selection, weight packing, workspaces
                outs = forward(i % n_slots)
                torch.cuda.synchronize()
                log("eager pass %d done" % i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if use_graph:
            try:
                for sl in range(n_slots):
                    graphs[sl] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graphs[sl]):
                        outs_slot[sl] = forward(sl)
                    graphs[sl].replay()
                    torch.cuda.synchronize()
                graph, outs = graphs[0], outs_slot[0]
            except Exception as e:                 # report, never hide: the JSON says graph=false
                if rank == 0:
                    print("[bench] hipGraph capture failed, running eager: %r" % (e,), file=sys.stderr)
                graph = None
                torch.cuda.synchronize()

        pending = [None] * n_slots
        step_no = [0]

        def step():
            nonlocal outs
            sl = step_no[0] % n_slots
            step_no[0] += 1
            if pending[sl] is not None:            # the collective that last read this slot's send buffer
                pending[sl].wait()
                pending[sl] = None
            if graph is not None:
                graphs[sl].replay()
                outs = outs_slot[sl]
            else:
                outs = forward(sl)
            if gather:
                pending[sl] = dg.gather(sl, async_op=True)

        def drain():
            for sl in range(n_slots):
                if pending[sl] is not None:
                    pending[sl].wait()
                    pending[sl] = None

        log("hipgraph=%s; warm-up" % (graph is not None))
        for _ in range(args.warmup):
            step()
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        log("timed %d steps: %.3f ms/step" % (args.steps, 1e3 * elapsed / args.steps))
        # frame 0 of what the LAST TIMED step produced (the replayed graph's static outputs): this is what the parity gate
        # below compares with the CPU oracle -- the timed outputs themselves, not a fresh forward
        timed_frame0 = [o[0:1].clone() for o in outs] if (rank == 0 and not args.decoder_only) else None
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())

        # ---- roofline leg: HIP events around every hand-written launch, eager, on the launch stream
        roof = None
        if rank == 0:
            tr = ops.KernelTrace()
            ops.set_trace(tr)
            nrep = max(3, min(args.steps, 10))
            # Kernel-quality leg: the forward once more, un-graphed, ONE launch at a time over the full per-GPU batch
            # (sub_batches = 1), HIP events around every hand-written launch.  The timed region above runs the very
            # same kernels as S concurrent sub-batches whose launches overlap on the chip, which is what buys the
            # wall-clock number but makes a per-launch duration there meaningless (rocprofv3 of the default command
            # shows the overlapped durations; `--streams 1` reproduces these isolated ones, see profiles/README.md).
            model.sub_batches = 1
            for _ in range(nrep):
                forward()
            model.sub_batches = S
            ops.set_trace(None)
            summ = tr.summary()
            # the dominant kernel = the hand-written kernel family with the most isolated time per step
            dom = max((k for k in summ if k.startswith("conv")), key=lambda k: summ[k]["ms"])
            d = summ[dom]
            algorithmic = d["flops"] / (d["ms"] * 1e-3) / 1e12
            executed = d["xflops"] / (d["ms"] * 1e-3) / 1e12
            # `achieved` / `frac` price the kernel by the FLOPs its own formulation issues to the matrix pipe.  For most
            # launches that IS the algorithmic count (2*M*c_out*c_in*taps with real channel counts); where the kernel
            # runs a cheaper exact formulation of the reference's op (sub-pixel upconv: 16 instead of 36 tap-products
            # per source pixel; tap-GEMM upconv: 9) the reference-formulation rate is reported next to it
            # (`algorithmic`) and may exceed the MFMA peak -- it is not a roofline position.
            achieved = min(algorithmic, executed)
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_MFMA_F32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_MFMA_F32_TFLOPS, 4),
                    "traffic": (pmc_traffic(dom) or {}).get("hbm_bytes_per_launch"),      # HBM bytes per launch (PMC) or null
                    "traffic_unit": "bytes/launch", "traffic_source": pmc_traffic(dom),
                    "executed": round(executed, 2), "executed_frac": round(executed / PEAK_MFMA_F32_TFLOPS, 4),
                    "algorithmic": round(algorithmic, 2),
                    "note": "achieved = FLOP the kernel issues to the MFMA pipe / HIP-event time of isolated full-batch "
                            "launches (the timed region overlaps %d sub-batches of the same kernels); algorithmic = the same "
                            "with the FLOP of the reference formulation (upconv as 3x3 on the upsampled map)" % S,
                    "launches_per_step": d["launches"] // nrep,
                    "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                    "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3)}
            # the same accounting per call-site group (tags) over every conv instantiation
            tags = {}
            for k, v in summ.items():
                for t, tv in v["tags"].items():
                    g = tags.setdefault(t, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, xflops=0.0))
                    for f in g:
                        g[f] += tv[f]
            groups = {}
            for t, g in sorted(tags.items()):
                e = {"launches_per_step": g["launches"] // nrep, "ms_per_step": round(g["ms"] / nrep, 3)}
                if g["flops"] > 0 and (t.startswith("enc") or t in ("aspp", "decoder_conv", "decoder_upconv")):
                    tf = g["flops"] / (g["ms"] * 1e-3) / 1e12
                    e.update(gflop_per_step=round(g["flops"] / nrep / 1e9, 2), achieved_tflops=round(tf, 2),
                             frac_mfma=round(tf / PEAK_MFMA_F32_TFLOPS, 4),
                             executed_tflops=round(g["xflops"] / (g["ms"] * 1e-3) / 1e12, 2))
                else:
                    gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9
                    e.update(gbs=round(gbs, 1), frac_hbm=round(gbs / PEAK_HBM_GBS, 4))
                    if t.startswith("reduc") and g["flops"] > 0:
                        # the 8x8 / 4x4 chains sit on the MFMA side of the ridge (AI 100 / 41 FLOP/B): both bounds
                        tf = g["flops"] / (g["ms"] * 1e-3) / 1e12
                        e.update(achieved_tflops=round(tf, 2), frac_mfma=round(tf / PEAK_MFMA_F32_TFLOPS, 4))
                groups[t] = e
            roof["groups"] = groups
            # SURVEY.md section 8(d)'s own roofline definitions for the hot path proper: ASPP = algorithmic FLOP of the five
            # branches / their time against the fp32-MFMA peak; reductions + LPG (fused: one launch per scale + reduc1x1) =
            # algorithmic bytes / their time against 8 TB/s
            def _t(name, key):
                return tags.get(name, {}).get(key, 0.0)
            if _t("aspp", "ms") > 0:
                tf = _t("aspp", "flops") / (_t("aspp", "ms") * 1e-3) / 1e12
                hp = {"aspp": {"gflop_per_step": round(_t("aspp", "flops") / nrep / 1e9, 2), "ms_per_step": round(_t("aspp", "ms") / nrep, 3),
                               "achieved_tflops": round(tf, 2), "frac_mfma": round(tf / PEAK_MFMA_F32_TFLOPS, 4),
                               "executed_tflops": round(_t("aspp", "xflops") / (_t("aspp", "ms") * 1e-3) / 1e12, 2)}}
                rms = _t("reduc", "ms") + _t("reduc_lpg", "ms")
                if rms > 0:
                    rb = _t("reduc", "bytes") + _t("reduc_lpg", "bytes")
                    gbs = rb / (rms * 1e-3) / 1e9
                    hp["reduction_lpg"] = {"mb_per_step": round(rb / nrep / 1e6, 1), "ms_per_step": round(rms / nrep, 3), "gbs": round(gbs, 1),
                                           "frac_hbm": round(gbs / PEAK_HBM_GBS, 4)}
                roof["survey_8d"] = hp
            kern = {}
            for k, v in summ.items():
                e = {"ms_per_step": round(v["ms"] / nrep, 4), "launches_per_step": v["launches"] // nrep}
                if k.startswith("conv"):
                    e["tflops"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)
                    e["executed_tflops"] = round(v["xflops"] / (v["ms"] * 1e-3) / 1e12, 2)
                else:
                    e["gbs"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
                    if k.startswith("reduc") and v["flops"] > 0:
                        e["tflops"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)
                kern[k] = e
            roof["kernels"] = kern
            conv_ms = sum(v["ms"] for k, v in summ.items() if k.startswith("conv"))
            conv_fl = sum(v["flops"] for k, v in summ.items() if k.startswith("conv"))
            roof["all_conv"] = {"ms_per_step": round(conv_ms / nrep, 3), "achieved_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2)}
            roof["hip_kernels_ms_per_step"] = round(sum(v["ms"] for v in summ.values()) / nrep, 3)

    if rank == 0:
        fps = G * args.steps / elapsed                       # whole-job frames (all ranks) over the slowest rank's time
        line = {
            "metric": "depth frames/sec at B=16, 352x1216 KITTI input",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BTS decoder hot path only" if args.decoder_only else "BtsModel.forward (encoder+decoder)")
                       + ", %s, B=%d per GPU, 3x%dx%d fp32%s" % (args.encoder, B, H, W,
                                                                  baseline_config_label(args.encoder, B, G, world, scaling, H, W)),
                       "batch_per_gpu": B, "global_batch": G, "image": "%dx%d" % (H, W),
                       "parallelism": "dp%d batch-sharded, RCCL weight broadcast once%s" % (
                           world, ", all-gather of 5 depth maps per step" if gather else ""),
                       "hipgraph": graph is not None, "sub_batch_streams": S, "fill_frames": fill_frames, "encoder_backend": args.encoder_backend,
                       "weights": "random-init encoder + PCG64(0) synthetic decoder"},
            "roofline": roof,
        }
        emu = None
        emu_frame0 = None
        if world == 1 and not args.no_emulated_leg:
            # Secondary measurement, NOT the headline: the same forward with every convolution in the
            # "fp32 emulated on the bf16 matrix cores" mode (bts_conv_desc.precision = 1: three-way bf16 split of both
            # operands, six products, fp32 accumulation -- results at fp32 rounding level, see its own parity gate).
            model.conv_precision = model.decoder.conv_precision = "bf16x3"
            try:
                with torch.no_grad():
                    forward()
                    torch.cuda.synchronize()
                    g2 = None
                    outs2 = [None]
                    if use_graph:
                        g2 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g2):
                            outs2[0] = forward()

                    def run2():
                        if g2 is not None:
                            g2.replay()
                        else:
                            outs2[0] = forward()
                    for _ in range(args.warmup):
                        run2()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        run2()
                    torch.cuda.synchronize()
                    el2 = time.perf_counter() - t1
                    if not args.decoder_only:
                        emu_frame0 = [o[0:1].clone() for o in outs2[0]]      # the timed outputs of this leg
                        torch.cuda.synchronize()
                emu = {"what": "same workload, convolutions in bts_conv_desc.precision=1 (fp32 products emulated with six "
                               "bf16 MFMAs per block after a three-way operand split, fp32 accumulate); secondary number, "
                               "the headline `value` uses the fp32-input MFMA",
                       "value": round(B * args.steps / el2, 3), "unit": "frames/s", "ms_per_step": round(1e3 * el2 / args.steps, 3)}
                log("emulated-fp32 (bf16x3) leg: %.3f ms/step" % (1e3 * el2 / args.steps))
            except Exception as e:       # a secondary measurement must never cost the headline line
                emu, emu_frame0 = None, None
                print("[bench] emulated-fp32 leg failed and is omitted: %r" % (e,), file=sys.stderr)
                torch.cuda.synchronize()
            finally:
                model.conv_precision = model.decoder.conv_precision = "fp32"
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle on host cores)")
            frame0 = timed_frame0             # outs[i][0:1] of the last timed step (graph replay or eager, as timed)
            line["cpu_baseline"], par, xpar = cpu_baseline(params, H, W, gpu_frame0=frame0,
                                                           extra_frame0={"bf16x3": emu_frame0} if emu_frame0 is not None else None)
            if par is not None:
                par["compared"] = "outs[i][0:1] of the last timed step (%s) vs the CPU oracle on the same image and weights" % (
                    "hipGraph replay" if graph is not None else "eager")
                line["parity"] = par
            if emu is not None and "bf16x3" in xpar:
                emu["parity"] = xpar["bf16x3"]
        if emu is not None:
            if "parity" in emu and not emu["parity"]["ok"]:
                emu["value"], emu["invalid"] = None, "parity gate of the emulated leg failed"
            line["emulated_fp32_bf16x3"] = emu
        parity_failed = "parity" in line and not line["parity"]["ok"]
        if parity_failed:                      # a fast kernel with wrong results is not a result: no clean-looking line
            line["parity_failed"] = True
            line["value_unverified"] = line["value"]
            line["value"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        if parity_failed:
            print("[bench] PARITY GATE FAILED: %r" % (line["parity"],), file=sys.stderr)
            if use_dist:
                dist.destroy_process_group()
            sys.exit(3)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
