#!/bin/bash
# Samples rocm-smi power / clocks while bench.py's timed region runs: is the step power-limited?
#   bash scripts/power_probe.sh <tag> [extra bench args]
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
python3 "$ROOT/bench.py" --steps 300 --warmup 5 --no-cpu-baseline --no-emulated-leg "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" &
BP=$!
sleep 6
for i in $(seq 1 14); do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' >> "$OUT/smi.txt"; echo >> "$OUT/smi.txt"
    sleep 0.5
done
wait $BP
tail -2 "$OUT/bench.err"
cat "$OUT/smi.txt" | cut -c1-400 | head -16
rocm-smi --showmaxpower 2>/dev/null | grep -i power
