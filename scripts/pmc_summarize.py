#!/usr/bin/env python3
"""Reduce a `rocprofv3 --kernel-trace --pmc ... --output-format csv` run to per-kernel ratios:
MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs), effective clock =
GRBM_GUI_ACTIVE/8 / duration, share of wave cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES), LDS-active fraction
(SQ_LDS_IDX_ACTIVE per CU-cycle), bank-conflict cycles.   python scripts/pmc_summarize.py <dir> [out.json]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    return re.sub(r"\(.*", "", name).split("::")[-1]


def main():
    d = sys.argv[1]
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            did = r.get("Dispatch_Id", r.get("Correlation_Id"))
            if did not in seen[k]:
                seen[k].add(did)
                agg[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for k, c in agg.items():
        if c["_ns"] <= 0 or "GRBM_GUI_ACTIVE" not in c:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # summed over 8 XCDs -> device cycles
        e = {"launches": len(seen[k]), "total_ms": round(c["_ns"] / 1e6, 3), "eff_clock_ghz": round(cyc / c["_ns"], 3)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
        if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
            e["wait_any_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
            e["wait_inst_any_frac"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        if "SQ_LDS_IDX_ACTIVE" in c:
            e["lds_active_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0), 4)
        if "SQ_LDS_BANK_CONFLICT" in c:
            e["lds_bank_conflict_cycles"] = int(c["SQ_LDS_BANK_CONFLICT"])
        if "SQ_ACTIVE_INST_VALU" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
            e["valu_issue_frac"] = round(c["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0), 4)
        out[k] = e
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
