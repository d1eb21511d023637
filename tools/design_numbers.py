"""The numbers DESIGN.md section 5 and profiles/README.md quote, printed FROM the committed JSON / text files of a round (so that the prose
cannot drift from them):   python tools/design_numbers.py r04"""
import json, os, re, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
load = lambda n: json.loads(open(os.path.join(P, n)).read().strip().splitlines()[-1]) if os.path.exists(os.path.join(P, n)) else None

b = load("%s_bench_line.json" % R)
if b:
    r = b["roofline"]
    print("bench line: %.1f images/s, %.3f ms/step, PCIe-inclusive %.1f; roofline.frac %.4f (achieved %.1f TFLOP/s, isolated %.1f), "
          "avg launch %.2f us (isolated %.2f), two-roof floor %.3f ms; groups: U-Net conv blocks %.3f ms = %.1f TFLOP/s = %.4f, detector_conv %.3f ms = %.1f TFLOP/s"
          % (b["value"], b["ms_per_step"], b.get("pcie_inclusive_images_per_s", float("nan")), r["frac"], r["achieved"], r["achieved_isolated"], r["avg_launch_us"],
             r["avg_launch_us_isolated"], r["two_roof_floor_ms"], r["groups"]["unet_conv_blocks_total"]["ms"], r["groups"]["unet_conv_blocks_total"]["tflops"],
             r["groups"]["unet_conv_blocks_total"]["frac"], r["groups"]["detector_conv"]["ms"], r["groups"]["detector_conv"]["tflops"]))
    c = b.get("cpu_baseline")
    if c:
        print("cpu_baseline: %s" % json.dumps({k: c[k] for k in c if k in ("value", "unit", "cores", "kind", "sample", "value_8_threads")}))
m = None
if os.path.exists(os.path.join(P, "%s_conv_mfma_util.json" % R)):
    m = json.load(open(os.path.join(P, "%s_conv_mfma_util.json" % R)))
    print("mfma_util (conv launches): %.4f; by kernel: %s" % (m["mfma_util"], ", ".join("%s %.4f (waiting %.3f)" % (k, v["mfma_util"], v.get("wave_time_waiting_cnt_or_barrier", float("nan")))
                                                                                       for k, v in m["by_kernel"].items())))
if os.path.exists(os.path.join(P, "%s_conv_traffic.json" % R)):
    t = json.load(open(os.path.join(P, "%s_conv_traffic.json" % R)))
    print("traffic: %.1f MB fetched + %.1f MB written = %.1f MB per launch over %s sampled launches; by kernel: %s; outside the figure: %s"
          % (t["fetch_bytes_per_launch"] / 1e6, t["write_bytes_per_launch"] / 1e6, t["traffic_bytes_per_launch"] / 1e6, t["launches_sampled"],
             json.dumps(t.get("by_kernel")), json.dumps(t.get("not_in_the_per_launch_figure"))))
ss = os.path.join(P, "%s_steady_state.txt" % R)
if os.path.exists(ss):
    txt = open(ss).read()
    print(re.search(r"steady state:.*", txt).group(0))
    for l in txt.splitlines():
        if re.match(r"\s{4}\S.*\s+[\d.]+\s+[\d.]+\s*$", l) or "everything else" in l:
            print(l)
