#!/bin/bash
# Copies the summaries of one scripts/collect_profiles.sh pass (gpurun_out/<tag>/) into profiles/ under the round's names.
#   bash scripts/install_profiles.sh r02e r02
set -e
TAG=$1; R=${2:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/$TAG; P=$ROOT/profiles
cp "$S/bench_n1.json" "$P/${R}_bench_n1.json"
cp "$S"/stats/runc/*_kernel_stats.csv "$P/${R}_bench_n1_kernel_stats.csv"
cp "$S"/stats_s1/runc/*_kernel_stats.csv "$P/${R}_bench_n1_streams1_kernel_stats.csv"
cp "$S/bench_under_rocprof.json" "$P/${R}_bench_n1_under_rocprof.json"
cp "$S/bench_streams1_under_rocprof.json" "$P/${R}_bench_n1_streams1_under_rocprof.json"
cp "$S/pmc_traffic.json" "$P/${R}_pmc_traffic.json"
cp "$S/pmc_mfma.json" "$P/${R}_pmc_mfma.json"
cp "$S/config3_bench.json" "$P/${R}_config3_resnext101_416x544_bench_n1.json"
cp "$S/shard_b8_bench.json" "$P/${R}_config3shard_b8_bench_n1.json"
cp "$S/train_step.txt" "$P/${R}_train_step_kernels.txt"
grep '^{' "$S/train_step.txt" | tail -1 > "$P/${R}_train_step.json"
for f in latency_b1 latency_b1_fill1 latency_b1_fill2 latency_b16; do [ -s "$S/$f.json" ] && cp "$S/$f.json" "$P/${R}_$f.json"; done
python3 - "$P" "$R" <<'PY'
import json, sys
P, R = sys.argv[1], sys.argv[2]
def L(f): return json.loads(open(f).read().strip().splitlines()[-1])
d = L("%s/%s_bench_n1.json" % (P, R))
print("bench %.1f frames/s %.3f ms parity=%s cpu=%.2f emu=%s fill=%s" % (d["value"], d["ms_per_step"], d["parity"]["ok"], d["cpu_baseline"]["value"],
      d.get("emulated_fp32_bf16x3", {}).get("value"), d["config"].get("fill_frames")))
r = d["roofline"]
print("dominant %s frac %.4f achieved %.2f avg_launch_us %.2f traffic %s" % (r["kernel"], r["frac"], r["achieved"], r["avg_launch_us"], r["traffic"]))
for f in ("config3_resnext101_416x544_bench_n1", "config3shard_b8_bench_n1", "bench_n1_streams1_under_rocprof", "bench_n1_under_rocprof"):
    x = L("%s/%s_%s.json" % (P, R, f)); print(f, x["value"], x["ms_per_step"])
print("train", json.load(open("%s/%s_train_step.json" % (P, R)))["ms_per_step"])
print("pmc meta", json.load(open("%s/%s_pmc_traffic.json" % (P, R)))["_meta"]["csrc_sha16"])
PY
