#!/bin/bash
# Copies what scripts/collect_profiles.sh <tag> left in gpurun_out/<tag> into profiles/<prefix>_* (the tracked, judged set).
#   bash scripts/publish_profiles.sh r02b r02
set -e
TAG=${1:?tag}; PFX=${2:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/$TAG; D=$ROOT/profiles
cp "$S/bench_n1.json" "$D/${PFX}_bench_n1.json"
cp "$S"/stats/*/*_kernel_stats.csv "$D/${PFX}_bench_n1_kernel_stats.csv"
cp "$S/bench_under_rocprof.json" "$D/${PFX}_bench_n1_under_rocprof.json"
cp "$S"/stats_s1/*/*_kernel_stats.csv "$D/${PFX}_bench_n1_streams1_kernel_stats.csv"
cp "$S/bench_streams1_under_rocprof.json" "$D/${PFX}_bench_n1_streams1_under_rocprof.json"
cp "$S/pmc_traffic.json" "$D/${PFX}_pmc_traffic.json"
cp "$S/pmc_mfma.json" "$D/${PFX}_pmc_mfma.json"
cp "$S/config3_bench.json" "$D/${PFX}_config3_resnext101_416x544_bench_n1.json"
cp "$S/shard_b8_bench.json" "$D/${PFX}_config3shard_b8_bench_n1.json"
tail -1 "$S/train_step.txt" > "$D/${PFX}_train_step.json"
grep -v "amdgpu.ids" "$S/train_step.err" > "$D/${PFX}_train_step_kernels.txt" || true
ls -la "$D" | grep "${PFX}_"
