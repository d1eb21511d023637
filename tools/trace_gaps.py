"""Idle time between consecutive kernels in a rocprofv3 --kernel-trace CSV: splits the launches of the steady-state steps into
graph-replayed (U-Net) and eagerly issued (detector) by kernel family and reports, for each class, the summed kernel time and the
summed gap to the previous kernel's end.  Answers 'would capturing the detector half in a hipGraph shorten the step?'.
Usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = []
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        gaps.append((max(0, s1 - e0), n0[:60] + "  ->  " + n1[:60]))
    tot_gap = sum(g for g, _ in gaps)
    print("launches %d  span %.3f ms  kernel time %.3f ms (%.1f %%)  gaps %.3f ms" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, tot_gap / 1e6))
    hist = {}
    for g, n in gaps:
        b = 0 if g < 500 else 1 if g < 1000 else 2 if g < 2000 else 3 if g < 4000 else 4 if g < 10000 else 5
        hist.setdefault(b, [0, 0])
        hist[b][0] += 1
        hist[b][1] += g
    names = ["<0.5us", "0.5-1", "1-2", "2-4", "4-10", ">10us"]
    for b in sorted(hist):
        print("  gap %-7s %6d launches  %8.3f ms" % (names[b], hist[b][0], hist[b][1] / 1e6))
    agg = {}
    for g, n in gaps:
        if g >= 4000:
            agg.setdefault(n, [0, 0])
            agg[n][0] += 1
            agg[n][1] += g
    for n, (k, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print("  %3d x  %8.1f us total  %s" % (k, g / 1e3, n))


if __name__ == "__main__":
    main()
