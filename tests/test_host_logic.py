"""CPU tests of the host-side mirror (feastkit.jl_amd): parameters, contours, partition and
the refinement loops, driven through a test-only oracle engine (tests/oracle_engine.py).
The product path itself has no CPU fallback -- see test_no_cpu_fallback."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import feast_oracle as fo
import feastkit_jl_amd as fk
from kat_util import cmat, cplx, load_kats, sparse_tridiag, tridiag
from oracle_engine import OracleEngine

K = load_kats()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fpm_sentinel_and_defaults():
    fpm = fk.feastinit()
    assert all(fpm[i] == fk.FEAST_UNINITIALIZED for i in range(1, 65))
    fpm[2] = 6                      # user override survives feastdefault!
    fk.feastdefault(fpm)
    assert fpm[2] == 6
    fpm = fk.feastdefault(fk.feastinit())
    for k, v in K["fpm_defaults"].items():
        assert fpm[int(k)] == v
    assert fpm[8] == 16 and fpm[16] == 0 and fpm[18] == 100
    assert fk.feast_tolerance(fpm) == 1e-12


@pytest.mark.parametrize("ne", [4, 6, 8, 12, 16, 24])
def test_contours_match_oracle(ne):
    fpm = fk.feastdefault(fk.feastinit())
    fpm[2] = ne
    fpm[8] = ne
    Z, W = fk.feast_contour(-0.3, 2.1, fpm)
    Zo, Wo = fo.feast_contour(-0.3, 2.1, ne)
    assert np.array_equal(Z, Zo) and np.array_equal(W, Wo)
    Zg, Wg = fk.feast_gcontour(0.5 + 0.25j, 1.7, fpm)
    Zgo, Wgo = fo.feast_gcontour(0.5 + 0.25j, 1.7, ne)
    assert np.array_equal(Zg, Zgo) and np.array_equal(Wg, Wgo)
    fpm[16] = 1
    assert np.array_equal(fk.feast_contour(0.0, 1.0, fpm)[1], fo.feast_contour(0.0, 1.0, ne, fpm16=1)[1])
    fpm[18], fpm[19] = 40, 30
    assert np.array_equal(fk.feast_gcontour(0.0, 1.0, fpm)[0], fo.feast_gcontour(0.0, 1.0, ne, 1, 40, 30)[0])
    assert fk.feast_inside_gcontour(0.2 + 0.1j, 0.0, 1.0, fpm) == fo.inside_gcontour(0.2 + 0.1j, 0.0, 1.0, 40, 30)


def test_distribute_contour_points_matches_reference_partition():
    for ne, nw in ((16, 8), (8, 3), (24, 4), (3, 5), (16, 1)):
        chunks = fo.distribute_contour_points(ne, nw)
        parts = fk.distribute_contour_points(ne, nw)
        assert [(c[0] if c else parts[i][0], len(c)) for i, c in enumerate(chunks)] == parts
        assert sum(n for _, n in parts) == ne


def test_balanced_partition_covers_every_node_once():
    for ne, nw in ((16, 8), (16, 4), (16, 3), (8, 2), (24, 4), (5, 8), (16, 1)):
        parts = fk.balanced_contour_points(ne, nw)
        assert sorted(e for p in parts for e in p) == list(range(ne))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert fk.balanced_contour_points(16, 8)[7] == [7, 8] and fk.balanced_contour_points(16, 8)[0] == [0, 15]


def test_input_checks_return_reference_error_codes():
    eng = OracleEngine()
    A = tridiag(5)
    fpm = fk.feastinit()
    assert fk.feast_hip_hermitian(eng, A, None, 1.0, 0.5, 3, fpm).info == 3     # Emin >= Emax
    assert fk.feast_hip_hermitian(eng, A, None, 0.0, 1.0, 0, fpm).info == 2     # M0 <= 0
    assert fk.feast_hip_hermitian(eng, A, None, 0.0, 1.0, 9, fpm).info == 2     # M0 > N
    assert fk.feast_hip_general(eng, A, None, 0.0, -1.0, 3, fpm).info == 4       # r <= 0
    with pytest.raises(ValueError):
        fk.feast(np.array([[1.0, 2.0, 0.0], [0.0, 3.0, 1.0], [0.5, 0.0, 4.0]]), np.eye(3), (0.5, 3.5), M0=3, engine=eng)
    with pytest.raises(ValueError):
        fk.feast(tridiag(3), None, (0.5, 3.5), M0=3, backend="threads", engine=eng)


def _variant_a_cases():
    k = K["tridiag3_real_sym"]
    yield "tridiag3", tridiag(3), np.eye(3), k["interval"], 3, {}, k["expect_lambda"]
    k = K["hermitian3_dense"]
    yield "herm3", cmat(k["A"]), None, k["interval"], 3, {}, k["expect_lambda"]
    k = K["hermitian3_sparse"]
    yield "herm3_sparse", sp.csc_matrix(cmat(k["A"])), None, k["interval"], 3, {}, k["expect_lambda"]
    k = K["diag80_oversized"]
    yield "diag80", np.diag(np.arange(1.0, 81)), None, k["interval"], 32, {2: 8, 3: 7, 4: 4}, k["expect_lambda"]
    k = K["tridiag10_backends"]
    yield "tridiag10", sparse_tridiag(10), None, k["interval"], 10, {2: 8, 4: 20}, k["expect_lambda"]
    k = K["hermitian_generalized_diag6"]
    yield ("diag6", sp.diags(np.array(k["A_diag"], dtype=complex)).tocsc(), sp.diags(np.array(k["B_diag"], dtype=complex)).tocsc(),
           k["interval"], 6, {}, k["expect_lambda"])


@pytest.mark.parametrize("case", list(_variant_a_cases()), ids=lambda c: c[0])
def test_hermitian_loop_equals_oracle_variant_a(case):
    """With the oracle engine and the complex half-contour sum the :hip host loop must walk
    the reference's variant A step for step: same M, loop count, info, eigenvalues."""
    name, A, B, interval, M0, fp, expect = case
    fpm = fk.feastinit()
    for i, v in fp.items():
        fpm[i] = v
    Q0 = fo.seeded_subspace(A.shape[0], M0)
    ref = fo.feast_hermitian(A, B, interval[0], interval[1], M0, ne=fp.get(2, 8), fpm3=fp.get(3, 12), fpm4=fp.get(4, 20), Q0=Q0)
    got = fk.feast_hip_hermitian(OracleEngine(), A, B, interval[0], interval[1], M0, fpm, solver="direct",
                                 real_projection=False, Q0=Q0)
    assert (got.info, got.M, got.loop) == (ref.info, ref.M, ref.loop)
    assert np.allclose(got.lambda_, ref.lam, atol=1e-12)
    assert np.allclose(np.sort(got.lambda_), sorted(expect), atol=1e-8)
    assert np.allclose(got.res, ref.res, atol=1e-10)


def test_real_projection_same_eigenpairs_fewer_loops():
    A, B, lam = fo.cfg3_problem(6, 5, 4)
    inside = lam[(lam >= 0) & (lam <= 2.0)]
    fpm = fk.feastinit()
    fpm[4] = 80
    half = fk.feast_hip_hermitian(OracleEngine(), A, B, 0.0, 2.0, len(inside) + 10, fpm, real_projection=False)
    full = fk.feast_hip_hermitian(OracleEngine(), A, B, 0.0, 2.0, len(inside) + 10, fk.feastinit(), real_projection=True)
    assert half.info == full.info == 0 and half.M == full.M == len(inside)
    assert np.allclose(np.sort(half.lambda_), inside, atol=1e-10) and np.allclose(np.sort(full.lambda_), inside, atol=1e-10)
    assert full.loop < half.loop / 3 and full.epsout <= 1e-12


def test_general_loop_equals_oracle_variant_c():
    k = K["general2"]
    A, B = cmat(k["A"]), cmat(k["B"])
    for Bin, expect in ((None, k["expect_standard"]), (B, k["expect_generalized"])):
        Q0 = fo.seeded_subspace(2, 2)
        ref = fo.feast_general(A, Bin, cplx(k["center"]), k["radius"], 2, Q0=Q0)
        got = fk.feast_hip_general(OracleEngine(), A, Bin, cplx(k["center"]), k["radius"], 2, fk.feastinit(), Q0=Q0)
        assert (got.info, got.M, got.loop) == (ref.info, ref.M, ref.loop)
        assert np.allclose(np.sort(got.lambda_.real), expect, atol=k["atol"])
    g = K["mpi_complex_general_diag4"]
    A = np.diag([cplx(v) for v in g["diag"]])
    fpm = fk.feastinit()
    fpm[3], fpm[4], fpm[8] = g["fpm3"], g["fpm4"], g["fpm8"]
    got = fk.feast_general(A, np.eye(4, dtype=complex), cplx(g["center"]), g["radius"], M0=4, fpm=fpm, engine=OracleEngine())
    want = sorted((cplx(v) for v in g["expect_lambda"]), key=lambda x: (x.real, x.imag))
    assert got.info == 0 and got.M == 3
    assert np.allclose(sorted(got.lambda_, key=lambda x: (round(x.real, 10), round(x.imag, 10))), want, atol=g["atol"])


def test_feast_api_real_input_returns_real_vectors():
    k = K["diag4_variant_b"]
    A = np.diag(k["diag"])
    fpm = fk.feastinit()
    fpm[2], fpm[4] = k["fpm2"], k["fpm4"]
    r = fk.feast(A, np.eye(4), tuple(k["interval"]), M0=4, fpm=fpm, engine=OracleEngine())
    assert r.info == 0 and r.M == k["expect_M"] and not np.iscomplexobj(r.q)
    assert np.allclose(np.sort(r.lambda_), k["expect_lambda"], atol=k["atol"])
    r2 = fk.feast(A, (0.4, 1.6), M0=4, fpm=fk.feastinit(), engine=OracleEngine())      # feast(A, interval) form
    assert r2.M == 3


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fk.FeastHipUnavailable):
        fk.HipEngine(0)
    with pytest.raises(fk.FeastHipUnavailable):
        fk.feast(tridiag(4), None, (0.5, 1.5), M0=2)


WORKER = r'''
import os, sys
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch.distributed as dist
import feast_oracle as fo, feastkit_jl_amd as fk
from oracle_engine import OracleEngine
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
A, B, lam = fo.cfg3_problem(6, 5, 4)
inside = lam[(lam >= 0) & (lam <= 2.0)]
fpm = fk.feastinit(); fpm[2] = 8
eng = OracleEngine()
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 2.0, len(inside) + 10, fpm, real_projection=True)
assert (eng.first, eng.count) == fk.distribute_contour_points(8, 2)[dist.get_rank()]
eng2 = OracleEngine()
rb = fk.feast_hip_hermitian(eng2, A, B, 0.0, 2.0, len(inside) + 10, fk.feastinit(), real_projection=True, node_assignment='balanced')
assert eng2.node_list == fk.balanced_contour_points(8, 2)[dist.get_rank()]
assert rb.M == r.M and np.allclose(np.sort(rb.lambda_), np.sort(r.lambda_), atol=1e-11)
eng3 = OracleEngine()
rc = fk.feast_hip_hermitian(eng3, A, B, 0.0, 2.0, 32, fk.feastinit(), real_projection=True, solver='bicgstab', warm_start=False, column_groups=2)
assert eng3.count == 8 and rc.info == 0 and rc.M == r.M and np.allclose(np.sort(rc.lambda_), np.sort(r.lambda_), atol=1e-10)
single = fk.feast_hip_hermitian(OracleEngine(), A, B, 0.0, 2.0, len(inside) + 10, fk.feastinit(), real_projection=True, group=dist.new_group([dist.get_rank()]) if False else None) if False else None
g = fk.feast_hip_general(OracleEngine(), np.diag([0.5+0.1j, 1.0+0.2j, 2.0-0.1j, 4.0]), None, 1.0+0.1j, 1.3, 4, fk.feastinit())
np.save(r"{out}/r%s.npy" % sys.argv[1], np.concatenate([[r.info, r.M, r.loop, r.epsout], np.sort(r.lambda_), [g.M], np.sort(g.lambda_.real)]))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_matches_single_process(tmp_path):
    """world_size-2 run of the sharded sweep (one all-reduce of Q_proj per loop) equals the
    single-process result: the N>1 path of bench.py, on gloo."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port, out=str(tmp_path)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)                     # every rank holds the same reduced result
    A, B, lam = fo.cfg3_problem(6, 5, 4)
    inside = lam[(lam >= 0) & (lam <= 2.0)]
    fpm = fk.feastinit(); fpm[2] = 8
    one = fk.feast_hip_hermitian(OracleEngine(), A, B, 0.0, 2.0, len(inside) + 10, fpm, real_projection=True)
    assert (int(r0[0]), int(r0[1]), int(r0[2])) == (one.info, one.M, one.loop)
    assert np.allclose(r0[4:4 + one.M], np.sort(one.lambda_), atol=1e-11)
    assert np.allclose(r0[4:4 + one.M], inside, atol=1e-10)
    assert int(r0[4 + one.M]) == 3


def test_variant_b_moments_loop_equals_oracle():
    """pfeast_hip_moments walks the reference's parallel 'moments' loop (feast_parallel.jl:450-572)."""
    k = K["diag4_variant_b"]
    A = np.diag(k["diag"]); B = np.eye(4)
    for Ain, Bin in ((A, B), (sp.csr_matrix(A), sp.identity(4, format="csr"))):
        fpm = fk.feastinit(); fpm[2], fpm[4] = k["fpm2"], k["fpm4"]
        Q0 = np.real(fo.seeded_subspace(4, 4))
        ref = fo.pfeast_moments(sp.csc_matrix(Ain) if sp.issparse(Ain) else Ain, sp.csc_matrix(Bin) if sp.issparse(Bin) else Bin,
                                *k["interval"], 4, ne=k["fpm2"], fpm4=k["fpm4"], Q0=Q0.copy())
        got = fk.pfeast_hip_moments(OracleEngine(), Ain, Bin, *k["interval"], 4, fpm, Q0=Q0.copy())
        assert (got.info, got.M, got.loop) == (ref.info, ref.M, ref.loop) == (0, 3, ref.loop)
        assert np.allclose(got.lambda_, ref.lam, atol=1e-12) and np.allclose(got.lambda_, k["expect_lambda"], atol=k["atol"])


# ---- complex-symmetric sibling (src/dense/feast_dense.jl:1026-1259) ---------------------------------
def _complex_symmetric_problem(n=24, seed=7, generalized=False):
    rng = np.random.default_rng(seed)
    d = np.linspace(-2.0, 2.0, n) + 1j * rng.uniform(-0.6, 0.6, n)
    G = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))          # real orthogonal: Qo^T = Qo^-1
    A = Qo @ np.diag(d) @ Qo.T + 0.0 * G
    A = 0.5 * (A + A.T)
    B = None
    if generalized:
        B = Qo @ np.diag(1.0 + 0.3 * rng.random(n)) @ Qo.T
        B = (0.5 * (B + B.T)).astype(complex)
    return A, B


@pytest.mark.parametrize("generalized", [False, True])
def test_complex_symmetric_matches_oracle(generalized):
    A, B = _complex_symmetric_problem(generalized=generalized)
    ev = np.linalg.eigvals(A if B is None else np.linalg.solve(B, A))
    c = 0.2 + 0.05j
    dist = np.sort(np.abs(ev - c))
    r = 0.5 * (dist[5] + dist[6])
    want = fo.feast_complex_symmetric(A, B, c, r, 10, ne=16, fpm3=10, fpm4=20)
    fpm = fk.feastinit(); fpm[8] = 16; fpm[3] = 10; fpm[4] = 20
    got = fk.feast_hip_complex_symmetric(OracleEngine(), A, B, c, r, 10, fpm)
    assert (got.info, got.M) == (want.info, want.M) == (0, 6)
    assert abs(got.loop - want.loop) <= 1
    inside = ev[np.abs(ev - c) <= r]
    key = lambda x: (round(x.real, 6), round(x.imag, 6))
    assert np.allclose(sorted(got.lambda_, key=key), sorted(inside, key=key), atol=1e-8)
    assert np.allclose(sorted(got.lambda_, key=key), sorted(want.lam, key=key), atol=1e-9)
    assert got.epsout <= 1e-10


def test_complex_symmetric_rejects_nonsymmetric():
    A = np.array([[1.0, 2.0], [0.0, 3.0]], dtype=complex)
    with pytest.raises(ValueError):
        fk.feast_hip_complex_symmetric(OracleEngine(), A, None, 0j, 1.0, 2, fk.feastinit())
    with pytest.raises(ValueError):
        fo.feast_complex_symmetric(A, None, 0j, 1.0, 2)


# ---- Zolotarev quadrature (fpm[16] = 2), src/core/feast_tools.jl:50-210, 263-266 ---------------------
def test_zolotarev_points_and_contour():
    x, w = fk.zolotarev_point(1, 1)
    assert x == 1j and abs(w - 0.99800399400799011j) < 1e-16            # table n = 1 (feast_tools.jl:51-53)
    _, w0 = fk.zolotarev_point(1, 0)
    assert abs(w0 + 0.49800399400799011) < 1e-16
    x, w = fk.zolotarev_point(3, 2)
    assert x == 1j and abs(w - 0.74467858236516826j) < 1e-16            # :60-63
    with pytest.warns(UserWarning):                                       # n not tabulated -> fallback rule (:196-209)
        x, w = fk.zolotarev_point(9, 1)
    assert abs(x - np.exp(1j * np.pi / 18)) < 1e-15 and abs(w - 1j * np.pi / 9) < 1e-15
    fpm = fk.feastdefault(fk.feastinit())
    fpm[2], fpm[16] = 8, 2
    Z, W = fk.feast_contour(1.0, 3.0, fpm)
    Zo, Wo = fo.feast_contour(1.0, 3.0, 8, fpm16=2)
    assert np.array_equal(Z, Zo) and np.array_equal(W, Wo)
    for e in range(8):
        x, w = fk.zolotarev_point(8, e + 1)
        assert Z[e] == x * 1.0 + 2.0 and W[e] == w * 1.0
    assert np.all(Z.imag > 0)                                             # upper half plane, symmetric pairs
    assert np.allclose(np.sort(Z.real - 2.0), -np.sort(Z.real - 2.0)[::-1])
    # the rational filter rho(lambda) = Re sum 2 w_e/(z_e - lambda) + we0 is ~1 inside, ~0 outside
    _, w0 = fk.zolotarev_point(8, 0)
    rho = lambda lam: float(np.real(np.sum(2 * W / (Z - lam))) + w0.real)
    assert abs(rho(2.0) - 1.0) < 0.05 and abs(rho(2.6) - 1.0) < 0.05
    assert abs(rho(5.0)) < 0.05 and abs(rho(-1.0)) < 0.05


def test_zolotarev_hermitian_solve_matches_oracle():
    A = tridiag(40)
    ev = 2 - 2 * np.cos(np.arange(1, 41) * np.pi / 41)
    lo, hi = 0.5 * (ev[7] + ev[8]), 0.5 * (ev[15] + ev[16])
    fpm = fk.feastinit(); fpm[2] = 8; fpm[16] = 2; fpm[4] = 40
    got = fk.feast_hip_hermitian(OracleEngine(), A, None, lo, hi, 12, fpm, real_projection=True)
    want = fo.feast_hermitian(A, None, lo, hi, 12, ne=8, fpm16=2, fpm4=40, real_projection=True)
    assert (got.info, got.M) == (want.info, want.M) == (0, 8)
    assert np.allclose(got.lambda_, ev[8:16], atol=1e-10) and np.allclose(got.lambda_, want.lam, atol=1e-10)


# ---- banded drivers (src/banded/feast_banded.jl) through the CPU stand-in engine ----------------------
def test_banded_drivers_host_logic():
    from feastkit_jl_amd import ingest
    import scipy.sparse as _sp
    n = 60
    rng = np.random.default_rng(13)
    o1 = 0.4 * (rng.standard_normal(n - 1) + 1j * rng.standard_normal(n - 1))
    H = _sp.diags([o1.conj(), np.linspace(1, 9, n), o1], [-1, 0, 1]).toarray()
    Hb = ingest.csr_to_band_upper(_sp.csr_matrix(H), 1)
    ev = np.linalg.eigvalsh(H)
    lo, hi = 0.5 * (ev[9] + ev[10]), 0.5 * (ev[15] + ev[16])
    fpm = fk.feastinit(); fpm[2] = 8; fpm[3] = 10; fpm[4] = 60
    r = fk.feast_hbev(Hb, 1, lo, hi, 10, fpm, engine=OracleEngine())
    assert r.info == 0 and r.M == 6 and np.allclose(r.lambda_, ev[10:16], atol=1e-8)
    with pytest.raises(ValueError):
        fk.feast_hbev(Hb, 1, lo, hi, 10, fk.feastinit(), engine=OracleEngine(), solver="cholesky")
    with pytest.raises(ValueError):
        fk.feast_sbev(np.zeros((1, 5)), 2, 0.0, 1.0, 2, engine=OracleEngine())


WORKER8 = r'''
import os, sys
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch.distributed as dist
import feast_oracle as fo, feastkit_jl_amd as fk
from oracle_engine import OracleEngine
rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=8)
A, B, lam = fo.cfg3_problem(8, 6, 5)
fpm = fk.feastinit(); fpm[2] = 16
eng = OracleEngine()
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 1.2, 64, fpm, real_projection=True, solver='bicgstab', warm_start=False,
                           node_assignment='balanced', column_groups='auto')
# 8 ranks, M0 = 64: 2 node groups x 4 column groups of 16 columns (the layout bench.py --gpus 8 takes)
assert eng.node_list == fk.balanced_contour_points(16, 2)[rank // 4], (rank, eng.node_list)
np.save(r"{out}/e%d.npy" % rank, np.concatenate([[r.info, r.M, r.loop, r.epsout], np.sort(r.lambda_)]))
dist.barrier(); dist.destroy_process_group()
'''


def test_eight_rank_gloo_bench_layout(tmp_path):
    """The N = 8 layout of bench.py (2 node groups x 4 column groups, balanced node lists) on gloo with the CPU
    engine mirror: all ranks agree and reproduce the single-process eigenvalues."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8.format(root=ROOT, port=port, out=str(tmp_path)))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(8)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [np.load(tmp_path / f"e{r}.npy") for r in range(8)]
    assert all(np.array_equal(res[0], x) for x in res[1:])
    A, B, lam = fo.cfg3_problem(8, 6, 5)
    inside = lam[(lam >= 0) & (lam <= 1.2)]
    fpm = fk.feastinit(); fpm[2] = 16
    one = fk.feast_hip_hermitian(OracleEngine(), A, B, 0.0, 1.2, 64, fpm, real_projection=True)
    assert (int(res[0][0]), int(res[0][1])) == (one.info, one.M) == (0, len(inside)) and abs(int(res[0][2]) - one.loop) <= 1
    assert np.allclose(res[0][4:4 + one.M], inside, atol=1e-10)


def test_feastdefault_mirrors_reference():
    """feastdefault! (src/core/feast_parameters.jl:41-386): defaults (KAT from test/runtests.jl: fpm[1..4] = 0, 8, 12, 20),
    the "<= 0" resets of fpm[2], fpm[4], fpm[8] (:103, :130, :161), fpm[30] untouched, range errors."""
    fpm = fk.feastdefault(fk.feastinit())
    for i, v in K["fpm_defaults"].items():
        assert fpm[int(i)] == v
    assert fpm[8] == 16 and fpm[18] == 100 and fpm[16] == 0 and fpm[30] == -111
    assert all(fpm[i] == 0 for i in list(range(20, 29)) + list(range(33, 36)) + list(range(50, 59)) + list(range(61, 64)))
    fpm = fk.feastinit(); fpm[2], fpm[4], fpm[8] = 0, -3, 0
    fk.feastdefault(fpm)
    assert (fpm[2], fpm[4], fpm[8]) == (8, 20, 16)
    fpm = fk.feastinit(); fpm[2] = 16; fpm[18] = 4000
    fk.feastdefault(fpm)
    assert fpm[2] == 16 and fpm[18] == 4000
    for slot, val in ((2, 21), (3, 17), (16, 3), (8, 1), (18, -1), (19, 181), (1, 2)):
        fpm = fk.feastinit(); fpm[slot] = val
        with pytest.raises(ValueError):
            fk.feastdefault(fpm)
    fpm = fk.feastinit(); fpm[2] = 24
    assert fk.feastdefault(fpm)[2] == 24            # allowed large Gauss rule


def test_hermitian_moments_driver_equals_oracle():
    """pfeast_hip_hermitian_moments walks _mpi_feast_complex_hermitian! (feast_mpi.jl:796-909) loop for loop: the
    reference fixture, and a genuinely complex Hermitian pencil (where the half-contour Q_proj is not the spectral
    projector -- the driver must still do exactly what the reference does)."""
    k = K["mpi_complex_hermitian_diag4"]
    A = np.diag(np.array(k["diag"], dtype=complex)); B = np.eye(4, dtype=complex)
    fpm = fk.feastinit(); fpm[2], fpm[4] = k["fpm2"], k["fpm4"]
    got = fk.pfeast_hip_hermitian_moments(OracleEngine(), A, B, *k["interval"], 4, fpm)
    want = fo.mpi_complex_hermitian(A, B, *k["interval"], 4, ne=k["fpm2"], fpm4=k["fpm4"])
    assert (got.info, got.M, got.loop) == (want.info, want.M, want.loop) == (0, 3, want.loop)
    assert np.allclose(got.lambda_, k["expect_lambda"], atol=k["atol"]) and np.allclose(got.lambda_, want.lam, atol=1e-12)
    n = 24
    rng = np.random.default_rng(11)
    H = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = np.diag(np.linspace(0.0, 6.0, n)) + 0.05 * (H + H.conj().T)
    G = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    B = np.eye(n) + 0.02 * (G + G.conj().T)
    ev = np.linalg.eigvalsh(np.linalg.solve(np.linalg.cholesky(B), np.linalg.solve(np.linalg.cholesky(B), A).conj().T))
    lo, hi = 0.5 * (ev[4] + ev[5]), 0.5 * (ev[11] + ev[12])
    for loops in (1, 3, 8):
        fpm = fk.feastinit(); fpm[2], fpm[3], fpm[4] = 8, 11, loops
        got = fk.pfeast_hip_hermitian_moments(OracleEngine(), A, B, lo, hi, 10, fpm)
        want = fo.mpi_complex_hermitian(A, B, lo, hi, 10, ne=8, fpm3=11, fpm4=loops)
        assert (got.info, got.M, got.loop) == (want.info, want.M, want.loop)
        assert np.allclose(got.lambda_, want.lam, atol=1e-9) and abs(got.epsout - want.epsout) <= 1e-6 * max(want.epsout, 1e-12) + 1e-13


def test_cost_balanced_contour_points():
    """Node lists from measured iteration counts: deterministic LPT, every node exactly once, heavy nodes apart."""
    costs = [170, 128, 91, 76, 59, 52, 49, 49, 49, 49, 56, 59, 78, 101, 165, 399]      # cfg 3 on the bench's contour
    groups = fk.cost_balanced_contour_points(costs, 2)
    assert sorted(groups[0] + groups[1]) == list(range(16))
    loads = [sum(costs[e] for e in g) for g in groups]
    assert abs(loads[0] - loads[1]) <= 0.05 * sum(costs)
    assert (15 in groups[0]) != (0 in groups[0]) or abs(loads[0] - loads[1]) < 60      # the two slowest nodes are not stacked
    assert fk.cost_balanced_contour_points(costs, 2) == groups                          # deterministic
    g3 = fk.cost_balanced_contour_points([5, 5, 5], 5)
    assert sorted(sum(g3, [])) == [0, 1, 2] and sum(1 for g in g3 if g) == 3
    g1 = fk.cost_balanced_contour_points([0, 0, 0, 0], 2)
    assert sorted(sum(g1, [])) == [0, 1, 2, 3] and all(len(g) == 2 for g in g1)


def test_split_balanced_assignment_covers_every_node_column_pair_once():
    """The heaviest nodes may be split by columns over k ranks; whatever the costs, every (node, column group) pair has
    exactly one owner, the layout is deterministic, and the largest share never exceeds the node-only LPT share."""
    from feastkit_jl_amd.contour import cost_balanced_contour_points, split_balanced_assignment
    cfg3 = [170, 128, 91, 76, 59, 52, 49, 49, 49, 49, 56, 59, 78, 101, 165, 399]        # measured node-iterations per step
    rng = np.random.default_rng(5)
    cases = [cfg3, [1] * 16, [1000] + [1] * 15, list(rng.integers(1, 500, 16)), list(rng.integers(1, 50, 7))]
    for costs in cases:
        for W in (1, 2, 3, 4, 6, 8):
            if W > len(costs):
                continue
            lay = split_balanced_assignment(costs, W)
            assert lay == split_balanced_assignment(list(costs), W) and len(lay) == W
            owners = {}
            for nodes, g, k in lay:
                assert 0 <= g < k and k in (1, 2, 4) and len(nodes) >= 1
                for e in nodes:
                    owners.setdefault(e, []).append((g, k))
            assert sorted(owners) == list(range(len(costs)))
            for e, parts in owners.items():
                k = parts[0][1]
                assert sorted(parts) == [(g, k) for g in range(k)]
            share = max(sum(costs[e] for e in nodes) / k for nodes, g, k in lay)
            base = max(sum(costs[e] for e in nodes) for nodes in cost_balanced_contour_points(costs, W))
            assert share <= base + 1e-9
    lay8 = split_balanced_assignment(cfg3, 8)
    assert lay8[0] == ([15], 0, 2) and lay8[1] == ([15], 1, 2)                          # node 15: 399 of 1630, fair share 204
    assert all(k == 1 for _n, _g, k in split_balanced_assignment(cfg3, 4))               # 399 ~ fair share of 4: no split


def test_contour_policy_filter_model():
    """The filter model of the policy, in the LIBRARY (feasthip_policy_filter_ratio / _reach through contour.filter_ratio /
    subspace_reach) against its numpy restatement (tests/policy_reference.py): the filter of the half contour with the real
    projection is ~1 inside, 1/2 at the interval ends, decays outside; the envelope ratio is invariant under shift and
    scaling of the interval, falls with the distance and rises with the ellipse ratio."""
    import policy_reference as pr
    from feastkit_jl_amd import contour as ct
    fpm = fk.feastinit(); fpm[2] = 16
    fk.feastdefault(fpm)
    Z, W = fk.feast_contour(2.0, 6.0, fpm)
    f = pr.filter_values(Z, W, np.array([2.0, 3.0, 4.0, 5.5, 6.0, 6.6, 8.0, 20.0, -3.0]))
    assert np.allclose(f[[1, 2, 3]], 1.0, atol=1e-6) and np.allclose(f[[0, 4]], 0.5, atol=1e-6)
    assert abs(f[5]) < 1e-3 and abs(f[6]) < 1e-8 and abs(f[7]) < 1e-12 and abs(f[8]) < 1e-10
    # library == restatement over ratios, quadratures, node counts, distances, with and without Ritz values inside
    rng = np.random.default_rng(1)
    for ne in (8, 16, 24):
        for q in (0, 1):
            for a in (100, 300, 1600, 4000, 8000):
                for d in (1.0, 1.2, 1.5, 2.5, 7.0):
                    ins = None if rng.random() < 0.5 else np.sort(rng.uniform(2.0, 6.0, 9))
                    lib, ref = ct.filter_ratio(2.0, 6.0, ne, q, a, d, ins), pr.filter_ratio(2.0, 6.0, ne, q, a, d, ins)
                    assert abs(lib - ref) <= 1e-6 * ref + 1e-14, (ne, q, a, d, lib, ref)
    # scale / shift invariance of the ratio, monotone in the distance, growing with the ellipse ratio
    for a in (100, 800, 4000):
        r1 = ct.filter_ratio(2.0, 6.0, 16, 0, a, 1.5)
        r2 = ct.filter_ratio(-0.3, 0.1, 16, 0, a, 1.5)
        assert abs(r1 / r2 - 1.0) < 1e-6
        assert ct.filter_ratio(2.0, 6.0, 16, 0, a, 1.2) >= r1 >= ct.filter_ratio(2.0, 6.0, 16, 0, a, 2.5)
    assert ct.filter_ratio(0, 1, 16, 0, 100, 1.5) < ct.filter_ratio(0, 1, 16, 0, 800, 1.5) < ct.filter_ratio(0, 1, 16, 0, 8000, 1.5)
    # the envelope is an upper bound of the filter beyond d
    fpm[18] = 2400
    Z, W = fk.feast_contour(0.0, 2.0, fpm)
    lam = 1.0 + np.linspace(1.5, 40.0, 3000)
    inn = np.abs(pr.filter_values(Z, W, np.linspace(0.0, 2.0, 65))).min()
    assert np.abs(pr.filter_values(Z, W, lam)).max() / inn <= ct.filter_ratio(0.0, 2.0, 16, 0, 2400, 1.5) * (1 + 1e-6)
    # subspace_reach: guards only, measured from the midpoint in half widths
    ritz = np.array([0.2, 0.5, 0.9, 1.3, 1.6, -0.4, 2.0])
    assert ct.subspace_reach(ritz, 0.0, 1.0, 1.0) == pytest.approx(3.0) == pr.subspace_reach(ritz, 0.0, 1.0, 1.0)
    assert ct.subspace_reach(ritz, 0.0, 1.0, 0.0) == pytest.approx(1.6) == pr.subspace_reach(ritz, 0.0, 1.0, 0.0)
    assert ct.subspace_reach(np.array([0.2, 0.5]), 0.0, 1.0) is None


def test_policy_state_machine_matches_reference():
    """feasthip_policy_init / _update / _set_aside (the C ABI's copy of the inexact-mode policy) against the numpy
    restatement over random trajectories: the same fpm[18], iteration cap and inner tolerance after every loop."""
    import ctypes as C
    import policy_reference as pr
    from feastkit_jl_amd import _lib
    lib = fk.load_library()
    rng = np.random.default_rng(7)
    for trial in range(40):
        Emin = float(rng.uniform(-2, 2)); Emax = Emin + float(rng.uniform(0.1, 3.0))
        ne, q = int(rng.choice([8, 12, 16])), int(rng.choice([0, 1]))
        rt, tol, cap0, steer = float(rng.choice([3e-2, 1e-2, 1e-1])), 1e-12, int(rng.choice([50, 100])), int(rng.random() < 0.8)
        pol = _lib.FeastHipPolicy()
        assert lib.feasthip_policy_init(C.byref(pol), Emin, Emax, ne, q, rt, tol, cap0, steer, 100) == 0
        ref = pr.PolicyReference(Emin, Emax, ne, q, rt, tol, cap0, steer, 100)
        assert pol.aspect == ref.aspect and pol.inner_cap == ref.inner_cap
        eps = float(rng.uniform(0.05, 1.0))
        r, mid = 0.5 * (Emax - Emin), 0.5 * (Emax + Emin)
        for loop in range(14):
            M = int(rng.integers(0, 20))
            inside = np.sort(rng.uniform(Emin, Emax, M))
            guards = mid + rng.choice([-1, 1], 12) * r * rng.uniform(1.05, 6.0 if loop < 3 else 2.0, 12)
            ritz = np.concatenate([inside, guards])
            capped = int(rng.random() < 0.3)
            eps *= float(rng.choice([0.03, 0.05, 0.2, 0.6, 0.9, 1.2]))
            rc = lib.feasthip_policy_update(C.byref(pol), eps, M, capped, ritz.ctypes.data_as(C.c_void_p), len(ritz))
            ref.update(eps, M, capped, ritz)
            assert rc == 0 and (pol.aspect, pol.inner_cap, pol.cap) == (ref.aspect, ref.inner_cap, ref.cap), (trial, loop)
            assert pol.next_rtol == pytest.approx(ref.next_rtol, rel=1e-14)
    # the last loop: 1.3e-12 for a target of 1e-12 relaxes the inner tolerance; 1.7e-11 does not
    pol = _lib.FeastHipPolicy()
    lib.feasthip_policy_init(C.byref(pol), 0.0, 1.0, 16, 0, 3e-2, 1e-12, 50, 0, 4000)
    z = np.zeros(1)
    lib.feasthip_policy_update(C.byref(pol), 1.7e-11, 0, 0, z.ctypes.data_as(C.c_void_p), 0)
    assert pol.next_rtol == 3e-2 and pol.aspect == 4000
    lib.feasthip_policy_update(C.byref(pol), 1.3e-12, 0, 0, z.ctypes.data_as(C.c_void_p), 0)
    assert 0.2 < pol.next_rtol <= 0.3
    # set-aside rule
    res = np.array([1e-6, 2e-6, 0.9, 3e-6, 0.4])
    flags = np.zeros(5, dtype=np.int32)
    assert lib.feasthip_policy_set_aside(res.ctypes.data_as(C.c_void_p), 5, flags.ctypes.data_as(C.c_void_p)) == 2 and flags.tolist() == [0, 0, 1, 0, 1]
    res = np.array([0.5, 0.9, 0.7])                     # nothing converged yet: nobody is 100x ahead
    assert lib.feasthip_policy_set_aside(res.ctypes.data_as(C.c_void_p), 3, flags.ctypes.data_as(C.c_void_p)) == 0 and flags[:3].tolist() == [0, 0, 0]


def test_contour_policy_driver_host_logic():
    """feast_hip_hermitian(contour_policy="auto") on the test engine: the policy engages only for inexact warm-started
    iterative solves with the real projection, leaves the caller's fpm untouched, records its choices, and the answer is
    the oracle's.  (The stand-in engine solves exactly, so this checks the control flow, not the speed.)"""
    import scipy.sparse as sp
    from oracle_engine import OracleEngine
    A, B, lam = fo.cfg3_problem(10, 8, 6)
    inside = lam[(lam >= 0.0) & (lam <= 0.8)]
    fpm = fk.feastinit(); fpm[2] = 8
    keep = fpm.copy()
    r = fk.feast_hip_hermitian(OracleEngine(), A, B, 0.0, 0.8, 24, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2,
                               solver_maxiter=100, contour_policy="auto")
    fk.feastdefault(keep)
    assert np.array_equal(fpm, keep)                                     # the caller's array only got its defaults
    assert r.info == 0 and r.M == len(inside) and np.abs(np.sort(r.lambda_) - inside).max() < 1e-10
    pol = r.stats["contour_policy"]
    assert len(pol["fpm18_per_loop"]) >= 1 and all(100 <= a <= 8000 for a in pol["fpm18_per_loop"]) and pol["fpm18_per_loop"][0] > 100
    # not engaged: direct solver / no warm start / explicit contour / complex Hermitian input
    for kw in (dict(solver="direct"), dict(solver="cocg", warm_start=False), dict(solver="cocg", warm_start=True, inner_rtol=3e-2,
               contour=fk.feast_contour(0.0, 0.8, keep))):
        r2 = fk.feast_hip_hermitian(OracleEngine(), A, B, 0.0, 0.8, 24, keep.copy(), contour_policy="auto", **kw)
        assert r2.info == 0 and r2.M == len(inside) and "contour_policy" not in r2.stats


def test_problem_fingerprint_is_position_dependent():
    """the fingerprint that lets set_problem() skip a re-ingest was a tuple of SUMS, so any permutation of
    values on one pattern collided and a reused engine silently solved the previous matrix.  It is a hash over the bytes of
    indptr / indices / data now; the three collisions the advisor reproduced must differ."""
    fp = fk.HipEngine._fingerprint
    # (i) A vs A^T for a structurally symmetric, non-symmetric convection-diffusion stencil (left eigenvectors of feast_general)
    n = 12
    A = sp.diags([-1.3 * np.ones(n - 1), 2.0 * np.ones(n), -0.7 * np.ones(n - 1)], [-1, 0, 1], format="csr")
    At = sp.csr_matrix(A.T)
    assert (A != At).nnz > 0 and np.array_equal(A.indices, At.indices) and np.array_equal(A.indptr, At.indptr)
    assert fp(A) != fp(At)
    # (ii) a tridiagonal Hamiltonian with two on-site energies swapped
    d = np.linspace(1.0, 2.0, n)
    H1 = sp.diags([-np.ones(n - 1), d, -np.ones(n - 1)], [-1, 0, 1], format="csr")
    d2 = d.copy(); d2[[3, 7]] = d2[[7, 3]]
    H2 = sp.diags([-np.ones(n - 1), d2, -np.ones(n - 1)], [-1, 0, 1], format="csr")
    assert fp(H1) != fp(H2)
    # (iii) binary alloy: two sites of a 0/1 potential swapped (same multiset of values, same pattern)
    v = np.array([0, 1] * (n // 2), dtype=float)
    v2 = v.copy(); v2[[0, 1]] = v2[[1, 0]]
    P1 = sp.diags([-np.ones(n - 1), 2 + v, -np.ones(n - 1)], [-1, 0, 1], format="csr")
    P2 = sp.diags([-np.ones(n - 1), 2 + v2, -np.ones(n - 1)], [-1, 0, 1], format="csr")
    assert fp(P1) != fp(P2)
    # equal content -> equal fingerprint (a copy, not the same object); None and non-CSR input keep their sentinels
    assert fp(A) == fp(A.copy()) and fp(None) is None and fp(A.tocsc()) is False and fp(A.toarray()) is False


def test_symmetric_kernel_sweeps_equal_oracle_rci():
    """feast_sbgv's driver (hip_backend.feast_hip_symmetric_kernel: one want_moments sweep per loop) against the oracle's
    restatement of what feast_srci! returns when its jobs are served (fo.rci_symmetric): standard and generalized."""
    n = 30
    S = tridiag(n)
    for Bm in (None, np.diag(np.linspace(1.0, 1.6, n))):
        fpm = fk.feastinit(); fpm[2] = 8; fpm[3] = 11; fpm[4] = 6
        got = fk.feast_hip_symmetric_kernel(OracleEngine(), S, Bm, 0.2, 1.3, 8, fpm)
        want = fo.rci_symmetric(S, Bm, 0.2, 1.3, 8, ne=8, fpm3=11, fpm4=6)
        assert (got.info, got.M, got.loop) == (want.info, want.M, want.loop)
        assert np.allclose(got.lambda_, want.lam, atol=1e-11) and np.allclose(got.res, want.res, rtol=1e-6, atol=1e-13)
        for j in range(got.M):
            assert min(np.linalg.norm(got.q[:, j] - want.q[:, j]), np.linalg.norm(got.q[:, j] + want.q[:, j])) <= 1e-8 * np.linalg.norm(want.q[:, j])


def test_direct_factors_are_released_when_the_call_returns():
    """band-LU factors outlived the call that made them (40 GB on cfg 3).  feast()/feast_general() release them on
    return unless keep_factors=True."""
    A = sp.csr_matrix(tridiag(24))
    for keep, want in ((False, 1), (True, 0)):
        eng = OracleEngine()
        r = fk.feast(A, None, (0.2, 1.3), M0=10, engine=eng, keep_factors=keep, real_projection=True)
        assert r.info == 0 and eng.calls.get("free_factors", 0) == want
        eng = OracleEngine()
        g = fk.feast_general(A.astype(complex), None, 0.8, 0.5, M0=8, engine=eng, keep_factors=keep)
        assert g.info == 0 and eng.calls.get("free_factors", 0) == want


def test_blas_limit_released_when_the_engine_raises():
    """an exception from the engine inside the refinement loop left the process-wide BLAS limit at one thread."""
    from feastkit_jl_amd.hip_backend import small_lapack

    class Boom(OracleEngine):
        def orthonormalize(self, *a, **k):
            raise fk.FeastHipError(7, "poisoned")
    with pytest.raises(fk.FeastHipError):
        fk.feast_hip_hermitian(Boom(), tridiag(20), None, 0.2, 1.3, 8, fk.feastinit())
    assert small_lapack._depth == 0 and small_lapack._outer is None
