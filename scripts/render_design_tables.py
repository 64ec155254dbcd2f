#!/usr/bin/env python3
"""Regenerates the per-group / per-kernel tables of DESIGN.md section 7 from profiles/<round>_bench_n1.json, so the document
never quotes numbers of an older build:   python scripts/render_design_tables.py [r02]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
d = json.loads(open(os.path.join(ROOT, "profiles", "%s_bench_n1.json" % R)).read().strip().splitlines()[-1])
r = d["roofline"]
out = ["| group | launches/step | ms/step (isolated B=16 launches) | algorithmic work/step | achieved |", "|---|---|---|---|---|"]
for k, v in sorted(r["groups"].items()):
    if "gflop_per_step" in v:
        ref = (", %.1f by the reference formulation's FLOPs" % v["achieved_tflops"]) if abs(v["achieved_tflops"] - v["executed_tflops"]) > 0.5 else ""
        out.append("| `%s` | %d | %.3f | %.1f GFLOP | %.1f TFLOP/s executed (%.0f %% of 157.3)%s |" % (
            k, v["launches_per_step"], v["ms_per_step"], v["gflop_per_step"], v["executed_tflops"], 100 * v["executed_tflops"] / 157.3, ref))
    else:
        extra = "; %.1f TFLOP/s (%.0f %% of the MFMA roof)" % (v["achieved_tflops"], 100 * v["frac_mfma"]) if "achieved_tflops" in v else ""
        out.append("| `%s` | %d | %.3f | — | %.0f GB/s (%.0f %% of 8 TB/s)%s |" % (k, v["launches_per_step"], v["ms_per_step"], v["gbs"], 100 * v["frac_hbm"], extra))
out += ["", "| kernel | launches/step | ms/step | rate |", "|---|---|---|---|"]
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if "executed_tflops" in v:
        ref = (", %.1f by the reference formulation's FLOPs" % v["tflops"]) if abs(v["tflops"] - v["executed_tflops"]) > 0.5 else ""
        out.append("| `%s` | %d | %.3f | %.1f TFLOP/s executed (%.2f of the MFMA roof)%s |" % (
            k, v["launches_per_step"], v["ms_per_step"], v["executed_tflops"], v["executed_tflops"] / 157.3, ref))
    else:
        out.append("| `%s` | %d | %.3f | %.0f GB/s (%.2f of 8 TB/s)%s |" % (
            k, v["launches_per_step"], v["ms_per_step"], v["gbs"], v["gbs"] / 8000, (", %.1f TFLOP/s" % v["tflops"]) if "tflops" in v else ""))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("| group | launches/step | ms/step (isolated B=16 launches)")
b = s.index("Notes.  `achieved` / `frac` in the bench line price")
open(p, "w").write(s[:a] + "\n".join(out) + "\n\n" + s[b:])
print("DESIGN.md tables rendered from %s_bench_n1.json: %.1f frames/s, %.3f ms/step, dominant %s frac %.4f traffic %s" % (
    R, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["traffic"]))
