#!/bin/bash
# bench.py under a list of extra command-line settings (perf loop tool): bash scripts/arg_sweep.sh OUTDIR "--streams 2" "--streams 8" ...
OUT=$1; shift
mkdir -p "$OUT"
i=0
for kv in "$@"; do
  i=$((i+1))
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-emulated-leg --steps 20 --warmup 5 $kv > "$OUT/a$i.json" 2> "$OUT/a$i.err"
  python3 - "$OUT/a$i.json" "$kv" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-60s %8.3f ms  %8.2f f/s" % (sys.argv[2], d["ms_per_step"], d["value"] or 0))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done
