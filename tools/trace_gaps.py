"""Idle time between consecutive kernels in a rocprofv3 --kernel-trace CSV: splits the launches of the steady-state steps into
graph-replayed (U-Net) and eagerly issued (detector) by kernel family and reports, for each class, the summed kernel time and the
summed gap to the previous kernel's end.  Answers 'would capturing the detector half in a hipGraph shorten the step?'.
Usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 0.5
    rows = []
    for r in csv.DictReader(open(path)):
        wg = 0
        try:
            wg = (int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))) * \
                 max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))) * max(1, int(r.get("Grid_Size_Z", 1)) // max(1, int(r.get("Workgroup_Size_Z", 1))))
        except (TypeError, ValueError):
            pass
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], wg))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    span = rows[-1][1] - rows[0][0]
    busy = sum(r_[1] - r_[0] for r_ in rows)
    gaps = []
    for (s0, e0, n0, _w0), (s1, e1, n1, _w1) in zip(rows, rows[1:]):
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
    # steady-state per-step table (the window holds whole steps only: from one adam_kernel to the last)
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    if len(ends) >= 2:
        win = rows[ends[0] + 1:ends[-1] + 1]
        nsteps = len(ends) - 1
        per = {}
        for s_, e_, n, _wg in win:
            k = n.replace("(anonymous namespace)::", "").replace("void ", "")
            per.setdefault(k, [0, 0])
            per[k][0] += 1
            per[k][1] += e_ - s_
        tot = sum(v[1] for v in per.values())
        aten = sum(v[1] for k, v in per.items() if "at::" in k or "rocclr" in k)
        naten = sum(v[0] for k, v in per.items() if "at::" in k or "rocclr" in k)
        print("steady state: %d steps, %.1f launches/step, kernel time %.3f ms/step, wall %.3f ms/step; ATen+copy %.1f launches, %.1f us per step"
              % (nsteps, len(win) / nsteps, tot / nsteps / 1e6, (win[-1][1] - win[0][0]) / nsteps / 1e6, naten / nsteps, aten / nsteps / 1e3))
        groups = [("hd_conv2d: conv_igemm (+ multi)", ("conv_igemm_kernel", "conv_igemm_multi_kernel")), ("hd_conv2d: conv3x3_w8", ("conv3x3_w8_kernel",)), ("hd_conv2d: conv3x3_m160 (producer / consumer)", ("conv3x3_m160_kernel",)), ("hd_conv2d: gemm_w8 (box head)", ("gemm_w8_kernel",)), ("hd_conv2d: register-resident (c64, stem, decoder block 3)", ("conv3x3_c64_kernel", "conv7x7s2_stem_kernel", "conv3x3_c32to128_kernel", "conv3x3_cat128to32_kernel")),
                  ("hd_conv2d: conv3x3_small", ("conv3x3_small_kernel",)), ("data + weight gradient, one grid", ("conv3x3_w8_wgrad_kernel",)),
                  ("weight gradients", ("::wgrad_kernel", "wgrad3x3_w8_kernel", "wgrad3x3_w8_multi_kernel", "wgrad3x3_small_kernel")), ("slab reductions", ("wgrad_reduce",)),
                  ("BatchNorm reduce (bwd)", ("bn_bwd_reduce",)), ("BatchNorm apply (bwd)", ("bn_bwd_apply",)), ("BatchNorm apply (fwd)", ("bn_apply_kernel",)),
                  ("BatchNorm finalize", ("bn_finalize",)), ("BatchNorm coefficients (bwd)", ("bn_bwd_coef",)),
                  ("RoIAlign", ("roi_align",)), ("NMS", ("nms_",)), ("Adam", ("adam_kernel",)), ("weight repack", ("weight_prep",)),
                  ("ATen + copies", ("at::", "rocclr"))]
        print("  groups (us/step, launches/step):")
        seen = 0.0
        for name, pats in groups:
            hit = lambda k: any((p_[2:] in k and "conv3x3_w8_wgrad" not in k) if p_.startswith("::") else p_ in k for p_ in pats)
            t = sum(v[1] for k, v in per.items() if hit(k))
            c = sum(v[0] for k, v in per.items() if hit(k))
            seen += t
            print("    %-32s %8.1f %7.1f" % (name, t / nsteps / 1e3, c / nsteps))
        print("    %-32s %8.1f" % ("everything else", (tot - seen) / nsteps / 1e3))
        if "--small" in sys.argv:          # one step's launches that cannot fill the chip: candidates for one-grid fusion with a neighbour
            one = rows[ends[0] + 1:ends[1] + 1]
            tot_small = 0.0
            for s_, e_, n, wg in one:
                if 0 < wg < 230 and e_ - s_ >= 6000:
                    tot_small += e_ - s_
                    print("  %4d blocks %7.1f us  %s" % (wg, (e_ - s_) / 1e3, n.replace("(anonymous namespace)::", "")[:110]))
            print("  launches with < 230 blocks and >= 6 us: %.1f us per step" % (tot_small / 1e3))
        if "--aten" in sys.argv:
            for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
                if "at::" in k or "rocclr" in k:
                    print("  %7.1f us/step %6.1f x %7.1f us  %s" % (t / nsteps / 1e3, c / nsteps, t / c / 1e3, k[:230]))
        if "--table" in sys.argv:
            for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
                print("  %7.1f us/step %6.1f x %7.1f us  %s" % (t / nsteps / 1e3, c / nsteps, t / c / 1e3, k[:100]))
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
