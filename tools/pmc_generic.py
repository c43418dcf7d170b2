"""Sum the counters of one or more rocprofv3 --pmc passes per kernel.
Usage: python tools/pmc_generic.py <out.json> "<command>" <counter_collection.csv> [more.csv ...]"""
import collections, csv, json, sys
acc = collections.OrderedDict()
for path in sys.argv[3:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0]
            d = acc.setdefault(k, collections.defaultdict(float))
            d[r["Counter_Name"]] += float(r["Counter_Value"])
out = {"command": sys.argv[2], "kernels": {}}
for k, d in acc.items():
    e = dict(d)
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e:
                e[c + "_over_WAVE_CYCLES"] = round(e[c] / wc, 4)
    if e.get("TCC_HIT_sum") is not None and e.get("TCC_MISS_sum") is not None and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
        e["L2_hit_rate"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
    out["kernels"][k] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, e in out["kernels"].items():
    print(k[:50], {x: e[x] for x in e if x.endswith("CYCLES") and "over" in x or x == "L2_hit_rate"})
