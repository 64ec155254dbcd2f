#!/usr/bin/env python3
"""HBM bytes per launch per kernel from two separate rocprofv3 PMC passes of the same command (FETCH_SIZE and
WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md): FETCH_SIZE is in KiB-like units of 1024 B? No -- rocprofv3
reports both in KB (1024 B); on gfx950 FETCH_SIZE counts wide streaming reads at half their bytes, so it is doubled
(MI355X_MICROARCH.md, HBM section).   python scripts/pmc_traffic.py <fetch_dir> <write_dir> out.json"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    m = re.search(r"(\w+)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    return re.sub(r"\(.*", "", name).split("::")[-1]


def collect(d, counter):
    tot, calls = defaultdict(float), defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            calls[k].add(r.get("Dispatch_Id", r.get("Correlation_Id")))
    return tot, {k: len(v) for k, v in calls.items()}


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, nw = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) & set(write)):
        if "conv" not in k and "reduc" not in k and "lpg" not in k and "pool" not in k and "get_depth" not in k \
                and "pack_planes" not in k and "nhwc" not in k:
            continue
        f = 2.0 * fetch[k] * 1024 / 1e6 / max(nf[k], 1)          # KB -> MB, x2 gfx950 wide-read correction
        w = write[k] * 1024 / 1e6 / max(nw[k], 1)
        out[k] = {"calls": nf[k], "fetch_MB_per_launch_x2": round(f, 3), "write_MB_per_launch": round(w, 3),
                  "hbm_MB_per_launch": round(f + w, 3)}
    from bts_amd import _lib
    commit = None
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass                                             # the GPU box snapshot has no .git: the source hash is the stamp
    out["_meta"] = {"csrc_sha16": _lib.source_hash(), "commit": commit,
                    "what": "HBM bytes per launch: 2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, separate --pmc passes"}
    json.dump(out, open(sys.argv[3], "w"), indent=0)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    main()
