"""bench.py's multi-rank plumbing, on CPU: `--gpus N` must start N ranks itself (the reference: mp.spawn(main_worker,
nprocs=ngpus), bts_main.py:843-847), report n_gpus = the ranks that really formed, and refuse -- non-zero exit, clear
message -- to run with fewer ranks than asked for.  `--launcher-selftest` swaps RCCL for gloo and skips the hot path
(which needs a GPU); everything else (launcher, rendezvous, shard plan, weight broadcast, barrier + max-over-ranks
timing, packed all-gather, the single JSON line on stdout) is the code the N-GPU run uses."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout                      # the contract: exactly ONE line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,scaling,per,total", [([], "weak", 16, 32), (["--global-batch", "64"], "strong", 32, 64)])
def test_gpus2_starts_two_ranks(extra, scaling, per, total):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"] + extra, env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["requested_gpus"] == 2
    assert (line["scaling"], line["batch_per_gpu"], line["global_batch"]) == (scaling, per, total)
    assert line["gather_ok"] and line["elapsed_is_max_over_ranks"] and line["shard"] == [0, per]


def test_gpus2_without_two_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node really has 2 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 requested" in r.stderr and r.stdout.strip() == ""      # no JSON line that could be mistaken for a result


def test_under_torchrun_world_size_must_match_gpus():
    port = _free_port()
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), BENCH, "--launcher-selftest"]
    ok = subprocess.run(base + ["--gpus", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    assert _one_json_line(ok.stdout)["n_gpus"] == 2
    bad = subprocess.run(base[:9] + [str(_free_port())] + base[10:] + ["--gpus", "1"], env=_env(), capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


@pytest.mark.parametrize("victim", [1, 0])
def test_one_dying_rank_stops_the_others_promptly(victim):
    """Only ONE rank fails (after the rendezvous, while the other is inside a collective): the parent must notice through
    its poll loop, terminate the survivor and return that rank's code -- not sit on rank 0's stdout until a collective
    times out (gloo's default: 30 minutes)."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"],
                       env=dict(_env(), BTS_BENCH_SELFTEST_FAIL_RANK=str(victim)), capture_output=True, text=True, timeout=240)
    assert r.returncode == 7, (r.returncode, r.stderr[-1500:])
    assert "rank %d exited with code 7" % victim in r.stderr
    assert r.stdout.strip() == ""                       # no JSON line from a run that lost a rank
    assert time.time() - t0 < 120


def test_uneven_global_batch_is_sharded_padded_and_trimmed():
    """7 frames over 2 ranks: 4 + 3 (DataParallel.scatter order); the gather pads the short rank and trims again
    (bts_amd.dist.DepthGather), so the line reports the real global batch and the gather check passes."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--global-batch", "7"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["global_batch"] == 7 and line["shard"] == [0, 4] and line["batch_per_gpu"] == 4 and line["gather_ok"]
    bad = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--global-batch", "1"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "smaller than" in bad.stderr


def test_kernel_labels_map_to_rocprof_names():
    """bench.py matches its own kernel labels to rocprofv3's demangled names (PMC traffic per launch): one pattern
    per kernel family, checked against names copied from a kernel-stats CSV."""
    import re
    sys.path.insert(0, ROOT)
    import bench
    seen = ["conv_fwd_kernel<64,128,2,4,32,false,0>", "conv_fwd_kernel<128,48,4,1,16,false,0>",
            "conv_halo_kernel<32,4,1,32,3,true,true,false>", "conv_halo_kernel<128,4,2,32,2,false,false,true>",
            "conv1x1_kernel<192,4,false>", "conv1x1_kernel<192,2,true>", "conv_halo_kernel<48,4,1,16,3,false,false,false>",
            "conv_halo_kernel<48,8,1,16,3,false,false,false>", "conv_halo_kernel<128,4,2,32,3,false,false,true,1>",
            "conv_halo_kernel<128,4,2,32,3,false,false,true,6>", "conv_halo_kernel<128,4,2,32,3,false,false,true,12>"]
    cases = {"conv_fwd_kernel<64,128,nhwc>": [seen[0]], "conv_fwd_kernel<128,48,nhwc,splitk>": [seen[1]],
             "conv_halo_kernel<32,k3,nchw,tail>": [seen[2]], "conv_halo_kernel<128,k2,nhwc>": [seen[3]],
             "conv1x1_kernel<192,4>": [seen[4]], "conv1x1_kernel<192,2>": [seen[5]],
             "conv_halo_kernel<48,k3,nhwc>": [seen[6]], "conv_halo_kernel<48,k3,nhwc,w8>": [seen[7]],
             "conv_halo_kernel<128,k3,nhwc>": [seen[8]], "conv_halo_kernel<128,k3,nhwc,dil>": [seen[9], seen[10]]}
    for label, want in cases.items():
        pat = bench.trace_to_rocprof_name(label)
        assert pat is not None, label
        assert [s for s in seen if re.fullmatch(pat, s)] == want, (label, pat)
    assert bench.trace_to_rocprof_name("get_depth_kernel<32>") is None
