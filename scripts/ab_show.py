#!/usr/bin/env python3
"""Print the A/B table of gpurun_out/<tag>/*.json bench lines: python scripts/ab_show.py <tag>"""
import glob, json, os, sys
pick = ["enc_b1_1x1", "enc_b2_1x1", "enc_b3_1x1", "enc_b3_3x3", "enc_b4_1x1", "enc_b4_3x3", "enc_stem", "aspp", "decoder_conv", "decoder_upconv"]
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    g = d["roofline"]["groups"]
    print("%-12s %7.2f ms %7.1f | " % (os.path.basename(f)[:-5], d["ms_per_step"], d["value"] or -1) +
          " ".join("%s=%.2f" % (k.replace("enc_", "").replace("decoder_", "d_"), g[k]["ms_per_step"]) for k in pick if k in g)
          + " | dom=%s frac=%.3f" % (d["roofline"]["kernel"], d["roofline"]["frac"]))
