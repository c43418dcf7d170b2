"""One-off larger runs of the driver-level fuzz tests (tests/test_gpu_driver_fuzz.py) with fresh seeds.
Usage: python tools/driver_fuzz_campaign.py <hermitian|wide|general|csym|inexact> [first_seed] [n_seeds] [cases_per_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import feastkit_jl_amd as fk
import test_gpu_driver_fuzz as t
which = sys.argv[1] if len(sys.argv) > 1 else "hermitian"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cases = int(sys.argv[4]) if len(sys.argv) > 4 else 8
eng = fk.HipEngine(0)
bad = 0
for seed in range(first, first + n):
    try:
        if which in ("hermitian", "wide"):
            t.test_driver_fuzz_vs_oracle(eng, seed, cases, which == "wide")
        elif which == "csym":
            t.test_complex_symmetric_driver_fuzz_vs_oracle(eng, seed, cases)
        elif which == "general":
            t.test_general_driver_fuzz_vs_oracle(eng, seed, cases)
        else:
            for prec, policy in ((64, None), (32, None), (64, "auto")):
                t.test_inexact_mode_fuzz(eng, seed, cases, prec, policy)
        print("seed", seed, "ok", flush=True)
    except AssertionError as ex:
        bad += 1
        print("seed", seed, "FAILED:", str(ex)[:400], flush=True)
print(f"campaign {which} done:", n * cases, "cases,", bad, "failing seeds")
sys.exit(1 if bad else 0)
