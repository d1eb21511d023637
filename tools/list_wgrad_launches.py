"""Durations of the multi-layer weight-gradient grids (M) and slab reductions (R) of ONE steady-state training step, in launch order,
from a rocprofv3 --kernel-trace CSV of tools/bench_step.py.
    python tools/list_wgrad_launches.py <kernel_trace.csv>"""
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends=[i for i,r in enumerate(rows) if "adam_kernel" in r[2]]
one=rows[ends[-3]+1:ends[-2]+1]
out=[]
for s,e,n in one:
    if "wgrad3x3_w8_multi" in n or "wgrad_reduce_multi" in n:
        out.append("%s %.1f" % ("M" if "w8_multi" in n else "R", (e-s)/1e3))
print(" ".join(out))
