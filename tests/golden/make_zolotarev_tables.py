"""Extract the Zolotarev quadrature constants (numbers only) that the reference tabulates in
src/core/feast_tools.jl:50-180 (FEAST libnum.f90, Guettel & Polizzi) into a JSON data file:
    {"n": {"we0": [re, im], "nodes": [[x_re, x_im, w_re, w_im], ...]}}
Run where /root/reference is mounted:  python tests/golden/make_zolotarev_tables.py [--product]

Two copies exist on purpose.  The default output is tests/golden/zolotarev_tables.json, the ORACLE's copy
(oracle/feast_oracle.py reads only that one); ``--product`` writes feastkit.jl_amd/zolotarev_tables.json, the file
the product ships.  The two are produced by different parsers of the same reference lines (a regular expression over
``complex(a, b)`` literals for the product copy, a line-by-line tokeniser for the oracle copy), and
tests/test_oracle_golden.py asserts that they hold the same numbers -- so a transcription slip in either extraction
shows up as a test failure instead of a silent agreement between oracle and product.
"""
import json
import os
import re
import sys

SRC = "/root/reference/src/core/feast_tools.jl"
HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT = "--product" in sys.argv
OUT = os.path.join(HERE, "..", "..", "feastkit.jl_amd", "zolotarev_tables.json") if PRODUCT else os.path.join(HERE, "zolotarev_tables.json")
num = r"(-?\d+\.\d*(?:[eE][-+]?\d+)?)"
cpx = re.compile(r"complex\(\s*" + num + r"\s*,\s*" + num + r"\s*\)")
text = open(SRC).read()
body = text[text.index("const ZOLOTAREV_TABLES"):text.index("function zolotarev_point")]


def tokenised_tables(body):
    """Second, independent extraction: walk the table line by line, track the current ``n => (`` entry, and read every
    ``complex(`` call by splitting on parentheses and commas (no regular expression over the numbers)."""
    out, cur, vals = {}, None, []

    def close():
        if cur is not None:
            we0, rest = vals[0], vals[1:]
            assert len(rest) == 2 * cur, (cur, len(rest))
            out[str(cur)] = {"we0": list(we0), "nodes": [[rest[2 * k][0], rest[2 * k][1], rest[2 * k + 1][0], rest[2 * k + 1][1]]
                                                          for k in range(cur)]}
    for line in body.splitlines():
        code = line.split("#", 1)[0]
        head = code.strip()
        if "=>" in head and head.split("=>")[0].strip().isdigit():
            close()
            cur, vals = int(head.split("=>")[0].strip()), []
            code = code.split("=>", 1)[1]
        if cur is None:
            continue
        parts = code.split("complex(")
        for piece in parts[1:]:
            a, b = piece.split(")", 1)[0].split(",")
            vals.append((float(a), float(b)))
    close()
    return out


tables = {}
for m in re.finditer(r"^\s*(\d+)\s*=>\s*\(", body, re.M):
    n = int(m.group(1))
    nxt = re.search(r"^\s*\d+\s*=>\s*\(", body[m.end():], re.M)
    chunk = body[m.end(): m.end() + nxt.start()] if nxt else body[m.end():]
    vals = [(float(a), float(b)) for a, b in cpx.findall(chunk)]
    we0, rest = vals[0], vals[1:]
    assert len(rest) == 2 * n, (n, len(rest))
    tables[str(n)] = {"we0": list(we0), "nodes": [[rest[2 * k][0], rest[2 * k][1], rest[2 * k + 1][0], rest[2 * k + 1][1]] for k in range(n)]}
if not PRODUCT:
    tables = tokenised_tables(body)
json.dump(tables, open(OUT, "w"), indent=0)
print("wrote", os.path.normpath(OUT), "n =", sorted(int(k) for k in tables))
