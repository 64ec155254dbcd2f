#!/usr/bin/env python3
"""Markdown tables for DESIGN.md section 7 from a bench.py JSON line:  python scripts/design_tables.py profiles/r02_final_bench_n1.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("| group | launches/step | ms/step (isolated B=16 launches) | algorithmic work/step | achieved |")
print("|---|---|---|---|---|")
for k, v in r["groups"].items():
    if "achieved_tflops" in v:
        print("| `%s` | %d | %.3f | %.1f GFLOP | %.1f TFLOP/s algorithmic (%.0f %% of 157.3), %.1f executed |"
              % (k, v["launches_per_step"], v["ms_per_step"], v["gflop_per_step"], v["achieved_tflops"], 100 * v["frac_mfma"], v["executed_tflops"]))
    else:
        print("| `%s` | %d | %.3f | — | %.0f GB/s (%.0f %% of 8 TB/s) |" % (k, v["launches_per_step"], v["ms_per_step"], v["gbs"], 100 * v["frac_hbm"]))
print()
print("| kernel | launches/step | ms/step | rate |")
print("|---|---|---|---|")
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    rate = ("%.1f TFLOP/s algorithmic, %.1f executed (%.2f of the MFMA roof)" % (v["tflops"], v["executed_tflops"], v["executed_tflops"] / 157.3)) if "tflops" in v \
        else "%.0f GB/s (%.2f of 8 TB/s)" % (v["gbs"], v["gbs"] / 8000.0)
    print("| `%s` | %d | %.3f | %s |" % (k, v["launches_per_step"], v["ms_per_step"], rate))
