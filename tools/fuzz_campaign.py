"""One-off larger run of the primitive fuzz (tests/test_gpu_fuzz.py) with fresh seeds.
Usage: python tools/fuzz_campaign.py [first_seed] [n_seeds] [trials_per_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import feastkit_jl_amd as fk
import test_gpu_fuzz as t
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 30
eng = fk.HipEngine(0)
bad = 0
for seed in range(first, first + n):
    try:
        t.test_primitive_fuzz(eng, seed, trials)
        print("seed", seed, "ok", flush=True)
    except AssertionError as ex:
        bad += 1
        print("seed", seed, "FAILED\n", str(ex)[:2000], flush=True)
print("campaign done:", n * trials, "trials,", bad, "failing seeds")
sys.exit(1 if bad else 0)
