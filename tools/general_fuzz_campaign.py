"""One-off larger run of the general-driver fuzz.  Usage: python tools/general_fuzz_campaign.py [first_seed] [n_seeds] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import feastkit_jl_amd as fk
import test_gpu_driver_fuzz as t
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng = fk.HipEngine(0)
bad = 0
for seed in range(first, first + n):
    try:
        t.test_general_driver_fuzz_vs_oracle(eng, seed, cases)
        print("seed", seed, "ok", flush=True)
    except AssertionError as ex:
        bad += 1
        print("seed", seed, "FAILED:", str(ex)[:600], flush=True)
print("campaign done:", n * cases, "cases,", bad, "failing seeds")
sys.exit(1 if bad else 0)
