"""Summarise the rocprofv3 --pmc passes written by tools/pmc_one_conv.sh: per conv kernel, mean counter values per launch and
the derived utilisations (MFMA busy / (SIMDs x wall clocks), LDS array busy, wave-time split)."""
import csv, glob, collections, sys, json
out = {}
for d in sorted(glob.glob('gpurun_out/pmc_*/*/*counter_collection.csv')):
    cfg = d.split('/')[1].split('_')[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k = r['Kernel_Name']
        if 'conv' not in k:
            continue
        agg[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        o = out.setdefault((cfg, k), {})
        for c, vals in v.items():
            o[c] = sum(vals) / len(vals)
for (cfg, k), o in out.items():
    print("W8=%s  %s" % (cfg, k))
    for c in sorted(o):
        print("    %-28s %14.0f" % (c, o[c]))
    if 'GRBM_GUI_ACTIVE' in o and 'SQ_VALU_MFMA_BUSY_CYCLES' in o:
        wall = o['GRBM_GUI_ACTIVE'] / 8
        print("    -> wall clocks %.0f ; MFMA pipe busy %.1f %% ; LDS array busy %.1f %% ; wave time: active %.0f %% / wait(cnt,barrier) %.0f %% / issue stall %.0f %%" % (
            wall, 100 * o['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / wall, 100 * o.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / wall,
            100 * o['SQ_ACTIVE_INST_ANY'] / o['SQ_WAVE_CYCLES'], 100 * o['SQ_WAIT_ANY'] / o['SQ_WAVE_CYCLES'], 100 * o['SQ_WAIT_INST_ANY'] / o['SQ_WAVE_CYCLES']))
