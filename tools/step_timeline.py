"""One steady-state step of a rocprofv3 --kernel-trace CSV as a timeline of its GAPS: every idle interval >= MIN us between two consecutive
kernels, with the time since the step began (the kernel after the previous step's adam_kernel) and the kernels on both sides.
    python tools/step_timeline.py <kernel_trace.csv> [min_gap_us=4] [step_index_from_end=2]"""
import csv
import sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
win = rows[lo:hi]
t0 = rows[lo - 1][1]          # end of the previous step's Adam
print("step of %d launches, %.3f ms from the previous Adam's end to this Adam's end; kernel time %.3f ms" % (
    len(win), (win[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in win) / 1e6))
prev_end, prev_name = t0, "adam_kernel (previous step)"
tot = 0.0
for i, (s, e, n) in enumerate(win):
    gap = (s - prev_end) / 1e3
    if gap >= min_gap:
        tot += gap
        print("  t = %8.1f us  gap %6.1f us  #%3d  %-60s -> %s" % ((prev_end - t0) / 1e3, gap, i, prev_name[:60], n[:70]))
    prev_end, prev_name = max(prev_end, e), n
print("gaps >= %.0f us: %.1f us in this step" % (min_gap, tot))
