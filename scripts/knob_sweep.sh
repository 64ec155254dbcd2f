#!/bin/bash
# bench.py under a list of environment-knob settings (perf loop tool): bash scripts/knob_sweep.sh OUTDIR "A=1 B=2" "A=3" ...
OUT=$1; shift
mkdir -p "$OUT"
i=0
for kv in "$@"; do
  i=$((i+1))
  ( export $kv; timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-emulated-leg --steps 20 --warmup 5 > "$OUT/s$i.json" 2> "$OUT/s$i.err" )
  python3 - "$OUT/s$i.json" "$kv" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-60s %8.3f ms  %8.2f f/s  parity=%s" % (sys.argv[2], d["ms_per_step"], d["value"] or 0, d.get("parity",{}).get("ok")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done
