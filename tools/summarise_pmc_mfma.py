#!/usr/bin/env python3
"""MFMA-pipe busy fraction per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (its own run,
--kernel-trace only): busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) -- the counter sums the
cycles of all SIMDs, GRBM_GUI_ACTIVE the active cycles of the 8 XCDs (round 1's definition, profiles/r01g_pmc_mfma.json).

    python tools/summarise_pmc_mfma.py gpurun_out/<tag>/pmc_mfma r04
"""
import collections, csv, glob, json, sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
out = {"note": "MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs), per kernel; "
               "rocprofv3 --pmc in its own pass (--kernel-trace only)", "kernels": {}}
for k, c in agg.items():
    gui, busy = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0 or busy <= 0:
        continue
    out["kernels"][k] = dict(launches=cnt[k], mfma_busy_frac=busy / (gui / 8.0 * 256 * 4),
                             gui_active_cycles_per_launch_per_xcd=gui / 8.0 / max(cnt[k], 1))
path = "profiles/%s_pmc_mfma.json" % sys.argv[2]
json.dump(out, open(path, "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["mfma_busy_frac"])[:8]:
    print("%-50s MFMA busy %.3f (%d launches)" % (k[:50], v["mfma_busy_frac"], v["launches"]))
print("wrote", path)
