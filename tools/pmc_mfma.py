"""Per-kernel MFMA utilisation from one rocprofv3 counter pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 <cmd>
MFMA busy fraction of a kernel = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (4 SIMDs * sum(SQ_BUSY_CU_CYCLES)) over its dispatches:
the matrix pipe of each of a CU's four SIMDs can be busy in every cycle the CU is busy (the gfx94x MfmaUtil formula
divides by GRBM_GUI_ACTIVE * CU count instead; both are reported).
Usage: python tools/pmc_mfma.py <counter_collection.csv> <out.json> "<command>" [n_cu]
"""
import collections
import csv
import json
import sys

acc = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0]
        d = acc.setdefault(k, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["_rows"] += 1
n_cu = int(sys.argv[4]) if len(sys.argv) > 4 else 256
out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "n_cu": n_cu, "kernels": {}}
for k, d in acc.items():
    mf, cu, gui = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("SQ_BUSY_CU_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
    if mf <= 0:
        continue
    ncount = max(1, sum(1 for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE") if c in d))
    out["kernels"][k] = {"dispatches": int(d["_rows"] / ncount),
                         "SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CU_CYCLES": cu, "GRBM_GUI_ACTIVE": gui,
                         "mfma_busy_over_4x_cu_busy": round(mf / (4 * cu), 4) if cu else None,
                         "mfma_util_gfx94x_formula": round(mf / (gui * n_cu * 4), 4) if gui else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items():
    print(k[:60], v["dispatches"], v["mfma_busy_over_4x_cu_busy"], v["mfma_util_gfx94x_formula"])
