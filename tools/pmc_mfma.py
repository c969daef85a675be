"""Fold a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) of the bench command into a per-kernel table.
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs); kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is the
                sum over the 8 XCDs; MI355X_MICROARCH.md 'DVFS give-back').  SQ_VALU_MFMA_BUSY_CYCLES counts cycles of MFMA-pipe
                occupancy summed over SIMDs (= 32 x N for v_mfma_f32_32x32x16_bf16, 16 x N for 16x16x32).
  clock_GHz   = kernel cycles / dispatch duration
  wait/issue/active shares = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycle units cancel)
usage: pmc_mfma.py <dir with *counter_collection.csv> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import kernel_key          # full name up to the parameter list: template instantiations stay apart

src, out = sys.argv[1], sys.argv[2]
disp = defaultdict(dict)
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            d = disp[(path, row["Dispatch_Id"])]
            d["name"] = row["Kernel_Name"]
            d["dur_ns"] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            d["vgpr"] = int(row["VGPR_Count"]) + int(row["Accum_VGPR_Count"])
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
acc = defaultdict(lambda: defaultdict(float))
for d in disp.values():
    a = acc[kernel_key(d["name"])]
    a["launches"] += 1
    for k, v in d.items():
        if k not in ("name",):
            a[k] += v
res = {}
for name, a in acc.items():
    n = a["launches"]
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    if cyc <= 0 or a["SQ_WAVE_CYCLES"] <= 0:
        continue
    res[name] = {
        "launches": int(n), "avg_us_profiled": round(a["dur_ns"] / n / 1e3, 1), "regs_per_lane": int(a["vgpr"] / n),
        "mfma_util": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4),
        "clock_GHz": round(cyc / a["dur_ns"], 3),
        "wave_wait_share": round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 3),
        "wave_issue_stall_share": round(a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 3),
        "wave_active_share": round(a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"], 3),
    }
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["avg_us_profiled"] * kv[1]["launches"]))
json.dump(res, open(out, "w"), indent=1)
for k, v in list(res.items())[:16]:
    print("%-92s n=%4d %8.1f us  mfma %.3f  clk %.2f  wait %.2f stall %.2f act %.2f" % (k, v["launches"], v["avg_us_profiled"], v["mfma_util"],
          v["clock_GHz"], v["wave_wait_share"], v["wave_issue_stall_share"], v["wave_active_share"]))
