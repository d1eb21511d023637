"""The largest idle gaps between consecutive kernels of the steady-state steps of a rocprofv3 --kernel-trace CSV, with the kernels on
either side -- where a step's wall time exceeds its kernel time.  Usage: python tools/trace_top_gaps.py <kernel_trace.csv> [min_us]"""
import collections
import csv
import sys

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
win = rows[ends[len(ends) // 2]:ends[-1] + 1]
nsteps = len(ends) - 1 - len(ends) // 2
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:70]
agg = collections.OrderedDict()
for (s0, e0, n0), (s1, e1, n1) in zip(win, win[1:]):
    g = (s1 - e0) / 1e3
    if g >= min_us:
        k = short(n0) + "  ->  " + short(n1)
        agg.setdefault(k, [0, 0.0])
        agg[k][0] += 1
        agg[k][1] += g
tot = sum(v[1] for v in agg.values())
print("%d steps; gaps >= %.1f us: %.1f us per step" % (nsteps, min_us, tot / nsteps))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%7.1f us/step  %4.1f x %6.1f us  %s" % (t / nsteps, n / nsteps, t / n, k))
