#!/bin/bash
# A/B runs of bench.py under environment knobs on a 1-GPU box; one line per variant into gpurun_out/$1/ab.txt
#   bash scripts/ab_bench.sh <tag> "NAME1:ENV1=.. ENV2=..:--extra --args" "NAME2:..."
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for spec in "$@"; do
    name=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; extra=${rest#*:}
    [ "$extra" == "$rest" ] && extra=""
    ( export $envs; timeout -k 10 200 python3 "$ROOT/bench.py" --no-cpu-baseline --no-emulated-leg $extra > "$OUT/$name.json" 2> "$OUT/$name.err" ) || { echo "$name FAILED rc=$?" | tee -a "$OUT/ab.txt"; tail -3 "$OUT/$name.err"; continue; }
    python3 - "$name" "$OUT/$name.json" <<'PY' | tee -a "$OUT/ab.txt"
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
g = d["roofline"]["groups"]
pick = ["enc_b1_1x1", "enc_b2_1x1", "enc_b3_1x1", "enc_b3_3x3", "enc_b4_1x1", "enc_b4_3x3", "enc_stem", "aspp", "decoder_conv", "decoder_upconv"]
print("%-14s %7.2f ms %7.1f f/s parity=%s | " % (sys.argv[1], d["ms_per_step"], d["value"] or -1, d.get("parity", {}).get("ok")) +
      " ".join("%s=%.2f" % (k.replace("enc_", "").replace("decoder_", "d_"), g[k]["ms_per_step"]) for k in pick if k in g))
PY
done
