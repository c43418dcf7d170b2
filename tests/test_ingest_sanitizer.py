"""CPU sanitizer build of the library's host-side ingest (sanitizers): the pure-C++ half of
feasthip_set_csr lives in feastkit.jl_amd/csrc/fh_ingest.hpp, which tests/host_ingest_harness.cpp compiles with gcc under
AddressSanitizer + UndefinedBehaviorSanitizer and fuzzes with random pencils in every input form the C ABI takes
(CSR/CSC, 0/1-based, unsorted rows, duplicates, empty rows, renumbering on and off, malformed pointers).  GPU
sanitizers are not available on this pool; the device side is covered by the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_ingest_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_ingest_harness"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
           os.path.join(ROOT, "tests", "host_ingest_harness.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert build.returncode == 0, build.stdout[-4000:]
    for seed in ("20260515", "7"):
        run = subprocess.run([str(exe), "400", seed], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
        assert run.returncode == 0 and run.stdout.strip().endswith("ok 400"), run.stdout[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_policy_under_asan_ubsan(tmp_path):
    """The inexact-mode host policy (csrc/fh_policy.hpp, exported as feasthip_policy_*) under the same sanitizers: quadrature
    nodes, filter model, reach and ratio choice over random and degenerate inputs (tests/host_policy_harness.cpp)."""
    exe = tmp_path / "host_policy_harness"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
           os.path.join(ROOT, "tests", "host_policy_harness.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert build.returncode == 0, build.stdout[-4000:]
    run = subprocess.run([str(exe), "150", "11"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0 and run.stdout.strip().endswith("ok 150"), run.stdout[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_multifrontal_plan_under_asan_ubsan(tmp_path):
    """The symbolic phase of the multifrontal sparse direct solver (csrc/fh_mf.hpp: nested dissection, fronts, padded groups,
    assembly and extend-add maps) under the same sanitizers: plan invariants on grid, random, disconnected and degenerate
    patterns, and the plan EXECUTED on the CPU (partial pivoting inside the fully-summed blocks, substitution through the
    tree) against the residual of the solve (tests/host_mf_harness.cpp); then the plan of cfg 3's pattern: a ninth of the
    band LU's work and under 0.75 GB of factors per quadrature node."""
    exe = tmp_path / "host_mf_harness"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
           os.path.join(ROOT, "tests", "host_mf_harness.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert build.returncode == 0, build.stdout[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout[-4000:]
    run = subprocess.run([str(exe), "stats", "50", "40", "25", "64"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout[-4000:]
    words = run.stdout.split()
    flops = float(words[words.index("flops") + 1])
    store = float(words[words.index("store") + 1])
    assert flops < 0.85e11 and store < 0.75, run.stdout
