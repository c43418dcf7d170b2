"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel busy time and the idle gaps between
consecutive kernels on the stream.  Usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
busy = collections.defaultdict(lambda: [0, 0])
gap_after = collections.defaultdict(lambda: [0, 0])
gap_list = collections.defaultdict(list)
tot_gap = 0
for i, (s, e, n) in enumerate(rows):
    k = n.split("(")[0][:60]
    busy[k][0] += e - s; busy[k][1] += 1
    if i + 1 < len(rows):
        g = rows[i + 1][0] - e
        if 0 < g < 2_000_000:       # ignore host-side pauses > 2 ms
            gap_after[k][0] += g; gap_after[k][1] += 1; gap_list[k].append(g)
            tot_gap += g
span = rows[-1][1] - rows[0][0]
print(f"span {span/1e6:.2f} ms, kernel busy {sum(v[0] for v in busy.values())/1e6:.2f} ms, small gaps {tot_gap/1e6:.2f} ms")
for k, (t, c) in sorted(busy.items(), key=lambda kv: -kv[1][0])[:14]:
    g, gc = gap_after[k]
    gl = sorted(gap_list[k]) or [0]
    print(f"{k:62s} n={c:6d} avg {t/c/1e3:8.1f} us   gap after: avg {g/max(gc,1)/1e3:6.1f} med {gl[len(gl)//2]/1e3:6.1f} p90 {gl[int(len(gl)*0.9)]/1e3:6.1f} max {gl[-1]/1e3:7.1f} us")
