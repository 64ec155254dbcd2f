#!/bin/bash
# Collects the round's measurement set on a 1-GPU box into gpurun_out/$1 (copy what should be judged into profiles/).
#   bash scripts/collect_profiles.sh r02g [r02]      (scratch tag, round prefix of the files under profiles/)
# Separate rocprofv3 passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass; no
# --pmc together with system traces).  The program after `--` is python3 itself (no env / bash -c hops).
set -o pipefail
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
echo "== rocprofv3 kernel stats, default command"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$B" --no-cpu-baseline --no-emulated-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || echo "stats failed"
echo "== rocprofv3 kernel stats, --streams 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_s1" -- python3 "$B" --streams 1 --no-cpu-baseline --no-emulated-leg > "$OUT/bench_streams1_under_rocprof.json" 2> "$OUT/stats_s1.err" || echo "stats s1 failed"
echo "== PMC FETCH_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-emulated-leg --streams 1 > /dev/null 2> "$OUT/pmc_fetch.err" || echo "fetch failed"
echo "== PMC WRITE_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-emulated-leg --streams 1 > /dev/null 2> "$OUT/pmc_write.err" || echo "write failed"
python3 "$ROOT/scripts/pmc_traffic.py" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.txt" 2>&1 || echo "pmc_traffic.py failed"
# the full bench line comes AFTER the counter passes: bench.py quotes roofline.traffic from profiles/<round>_pmc_traffic.json,
# and only while that file carries the hash of the kernel sources in the tree -- i.e. this pass's own measurement
ROUND=${2:-r02}
[ -s "$OUT/pmc_traffic.json" ] && cp "$OUT/pmc_traffic.json" "$ROOT/profiles/${ROUND}_pmc_traffic.json"
echo "== bench (full line: cpu baseline, parity gate, emulated leg, PMC traffic of this pass)"
timeout -k 10 400 python3 "$B" --steps 20 --warmup 5 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || echo "bench failed rc=$?"
tail -1 "$OUT/bench_n1.err"
echo "== PMC MFMA / waits"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 "$B" --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-emulated-leg > /dev/null 2> "$OUT/pmc_mfma.err" || echo "mfma failed"
python3 "$ROOT/scripts/pmc_summarize.py" "$OUT/pmc_mfma" "$OUT/pmc_mfma.json" > /dev/null 2>&1 || echo "pmc_summarize failed"
# raw counter dumps are large: keep the summaries and the stats CSVs only
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
echo "== config 3 (ResNeXt101, 416x544)"
timeout -k 10 300 python3 "$B" --encoder resnext101_bts --dataset nyu --height 416 --width 544 --no-emulated-leg > "$OUT/config3_bench.json" 2> "$OUT/config3.err" || echo "config3 failed"
echo "== config 5 per-GPU workload (training step B=4 352x704)"
timeout -k 10 300 python3 "$ROOT/scripts/train_bench.py" --trace --steps 5 --warmup 3 > "$OUT/train_step.txt" 2> "$OUT/train_step.err" || echo "train failed"
echo "== config 3 strong-scaling shard (B=64 on one GPU is configs[3]'s whole batch; here the 8-frame rank shard)"
timeout -k 10 300 python3 "$B" --batch 8 --no-cpu-baseline --no-emulated-leg > "$OUT/shard_b8_bench.json" 2> "$OUT/shard_b8.err" || echo "b8 failed"
ls -la "$OUT"
