"""Combine two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE
pass of the same command) into per-kernel HBM bytes per launch.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half of a wide coalesced
read -> bytes = 2 * FETCH_KB * 1024; WRITE_SIZE is exact -> bytes = WRITE_KB * 1024.
Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command>" [kernel_source_hash]
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4] if len(sys.argv) > 4 else "",
       "kernel_source_hash": sys.argv[5] if len(sys.argv) > 5 else "",      # bench.py quotes the traffic only for matching kernel sources
       "correction": "gfx950: FETCH_SIZE counts half of a wide coalesced read (MI355X_MICROARCH.md, HBM): bytes = 2*FETCH_KB*1024; "
                     "WRITE_SIZE exact: bytes = WRITE_KB*1024",
       "kernels": {}}
for k, (fs, n) in fetch.items():
    ws, wn = write.get(k, (0.0, 0))
    mf, mw = fs / max(n, 1), ws / max(wn, 1)
    out["kernels"][k] = {"launches": n, "mean_FETCH_KB": round(mf, 1), "mean_WRITE_KB": round(mw, 1),
                         "mean_hbm_bytes_per_launch": int(2 * mf * 1024 + mw * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], "kernels:", len(out["kernels"]))
