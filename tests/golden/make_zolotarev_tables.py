"""Extract the Zolotarev quadrature constants (numbers only) that the reference tabulates in
src/core/feast_tools.jl:50-180 (FEAST libnum.f90, Guettel & Polizzi) into a JSON data file:
    {"n": {"we0": [re, im], "nodes": [[x_re, x_im, w_re, w_im], ...]}}
Run where /root/reference is mounted:  python tests/golden/make_zolotarev_tables.py
"""
import json
import os
import re

SRC = "/root/reference/src/core/feast_tools.jl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "feastkit.jl_amd", "zolotarev_tables.json")
num = r"(-?\d+\.\d*(?:[eE][-+]?\d+)?)"
cpx = re.compile(r"complex\(\s*" + num + r"\s*,\s*" + num + r"\s*\)")
text = open(SRC).read()
body = text[text.index("const ZOLOTAREV_TABLES"):text.index("function zolotarev_point")]
tables = {}
for m in re.finditer(r"^\s*(\d+)\s*=>\s*\(", body, re.M):
    n = int(m.group(1))
    nxt = re.search(r"^\s*\d+\s*=>\s*\(", body[m.end():], re.M)
    chunk = body[m.end(): m.end() + nxt.start()] if nxt else body[m.end():]
    vals = [(float(a), float(b)) for a, b in cpx.findall(chunk)]
    we0, rest = vals[0], vals[1:]
    assert len(rest) == 2 * n, (n, len(rest))
    tables[str(n)] = {"we0": list(we0), "nodes": [[rest[2 * k][0], rest[2 * k][1], rest[2 * k + 1][0], rest[2 * k + 1][1]] for k in range(n)]}
json.dump(tables, open(OUT, "w"), indent=0)
print("wrote", os.path.normpath(OUT), "n =", sorted(int(k) for k in tables))
