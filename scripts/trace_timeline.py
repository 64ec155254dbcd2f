#!/usr/bin/env python3
"""Where the time of a multi-stream step goes: rocprofv3 --kernel-trace CSV -> per-kernel-family attributed time.
At every instant the wall clock is split equally among the kernels in flight (sweep line); also reports how long
1, 2, 3, 4+ kernels overlapped and the idle gaps.   python scripts/trace_timeline.py <kernel_trace.csv> [t0_frac t1_frac]"""
import csv, re, sys, collections

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"(\w+)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else n.split("(")[0]

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Queue_Id"]),
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
T0, T1 = rows[0][0], max(r[1] for r in rows)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lo, hi = T0 + f0 * (T1 - T0), T0 + f1 * (T1 - T0)
ev = []
for s, e, n, q, g in rows:
    if e <= lo or s >= hi: continue
    ev.append((max(s, lo), 1, n)); ev.append((min(e, hi), -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
active = collections.Counter(); attributed = collections.Counter(); conc = collections.Counter(); busy = collections.Counter()
prev = lo; nact = 0
for t, d, n in ev:
    dt = t - prev
    if dt > 0:
        conc[min(nact, 5)] += dt
        if nact:
            for k, c in active.items():
                if c: attributed[k] += dt * c / nact; busy[k] += dt * c
    prev = t
    active[n] += d; nact += d
tot = hi - lo
print("window %.3f ms; overlap histogram (kernels in flight: share of wall): %s" % (tot / 1e6, {k: round(v / tot, 3) for k, v in sorted(conc.items())}))
print("%-60s %10s %10s" % ("kernel", "attrib ms", "sum-dur ms"))
for k, v in attributed.most_common(40):
    print("%-60s %10.3f %10.3f" % (k[:60], v / 1e6, busy[k] / 1e6))
