"""MFMA-pipe utilisation per conv kernel from rocprofv3 PMC passes of a training-step run.

    STEPS=3 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        --kernel-trace -d gpurun_out/pmc_mfma --output-format csv -- python3 tools/bench_step.py
    STEPS=3 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmc_gui ... (second pass)
    python tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/pmc_gui profiles/r02_conv_mfma_util.json

mfma_util of a kernel = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum wall clocks), wall clocks per launch = SQ_BUSY_CYCLES / 32
(summed over the 32 shader engines; it reproduces launch duration x ~1.98 GHz here, whereas GRBM_GUI_ACTIVE / 8 reads ~45 % high on
these 10-50 us dispatches, as MI355X_MICROARCH.md "DVFS give-back" warns for dispatches under 0.3 ms -- it is kept as `gui_clocks`).  SQ_VALU_MFMA_BUSY_CYCLES counts clocks (32 per
v_mfma_f32_32x32x16_f16), SQ_WAVE_CYCLES / SQ_WAIT_* count quad-clocks.  Kernels: every template instance of the conv families
(conv_igemm_kernel, conv3x3_w8_kernel, conv3x3_m160_kernel, conv3x3_c64_kernel, conv3x3_small_kernel) and the weight-gradient kernels."""
import collections
import csv
import glob
import json
import sys

FAMILIES = ["conv_igemm_kernel", "gemm_w8_kernel", "conv3x3_w8_kernel", "conv3x3_m160_kernel", "conv3x3_c64_kernel", "conv7x7s2_stem_kernel", "conv3x3_c32to128_kernel", "conv3x3_cat128to32_kernel", "conv3x3_small_kernel", "wgrad3x3_w8_multi_kernel", "wgrad3x3_w8_kernel", "wgrad_kernel", "wgrad3x3_small_kernel"]


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            fam = next((k for k in FAMILIES if k in r["Kernel_Name"]), None)
            if fam is None:
                continue
            acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            n[fam][r["Counter_Name"]] += 1
    return acc, n


a, na = load(sys.argv[1])
b, nb = load(sys.argv[2])
out = {"by_kernel": {}, "command": "STEPS=3 rocprofv3 --pmc <SQ counters | GRBM_GUI_ACTIVE ...> --kernel-trace -- python3 tools/bench_step.py (two passes)",
       "definition": "mfma_util = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum SQ_BUSY_CYCLES / 32); x (clock / 2.4 GHz) bounds the fraction of the 2.5 PFLOP/s roof"}
busy_all = wall_all = 0.0
for fam in FAMILIES:
    if fam not in a or fam not in b or not b[fam].get("GRBM_GUI_ACTIVE"):
        continue
    # the two passes see the same launches: scale by launch counts in case they differ
    launches_a, launches_b = na[fam]["SQ_VALU_MFMA_BUSY_CYCLES"], nb[fam]["GRBM_GUI_ACTIVE"]
    wall = a[fam]["SQ_BUSY_CYCLES"] / 32.0
    busy = a[fam]["SQ_VALU_MFMA_BUSY_CYCLES"]
    e = {"launches": launches_a, "mfma_util": round(busy / 1024.0 / wall, 4), "wall_clocks_per_launch": round(wall / launches_a),
         "gui_clocks_per_launch": round(b[fam]["GRBM_GUI_ACTIVE"] / 8.0 / max(launches_b, 1))}
    wc = a[fam].get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        e["wave_time_waiting_cnt_or_barrier"] = round(a[fam].get("SQ_WAIT_ANY", 0.0) / wc, 3)
        e["wave_time_issue_stall"] = round(a[fam].get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
        e["wave_time_lds_issue_stall"] = round(a[fam].get("SQ_WAIT_INST_LDS", 0.0) / wc, 3)
    if a[fam].get("SQ_LDS_IDX_ACTIVE"):
        e["lds_array_busy"] = round(a[fam]["SQ_LDS_IDX_ACTIVE"] / 256.0 / wall, 4)
        e["lds_bank_conflict_share"] = round(a[fam].get("SQ_LDS_BANK_CONFLICT", 0.0) / a[fam]["SQ_LDS_IDX_ACTIVE"], 4)
    out["by_kernel"][fam] = e
    if fam.startswith("conv"):
        busy_all += busy
        wall_all += wall
out["mfma_util"] = round(busy_all / 1024.0 / wall_all, 4) if wall_all else None
out["commit"] = __import__("os").environ.get("HD_COMMIT")
sys_path_ = __import__("sys").path; sys_path_.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
out["csrc_digest"] = __import__("hallucidet_amd.build", fromlist=["source_digest"]).source_digest()   # bench.py flags the summary as stale on any other build      # the tree this profile was taken on (the GPU box has no .git)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
