"""The collective inside the C ABI (include/feasthip.h: feasthip_comm_*): the image of MPI.Allreduce in
src/parallel/feast_mpi.jl:117-119, 856-858 and of the master sum src/parallel/feast_parallel.jl:497-503.

RCCL refuses two ranks of one communicator on the same HIP device, and the test box has one GPU: the RCCL
transport is exercised with a single rank, the multi-rank logic (packed reduce inside contour_apply, global
node status, column blocks, failure propagation) through the shared-device transport, driven from plain
ctypes workers that exchange the unique id through a file -- no torch.distributed anywhere."""
import os
import subprocess
import sys

import numpy as np
import pytest

import feast_oracle as fo
import feastkit_jl_amd as fk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_single_rank(engine):
    import torch
    eng = fk.HipEngine(0)
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(1, 0, uid, "rccl")
    assert (eng.comm_size, eng.comm_rank, eng.comm_transport()) == (1, 0, 1)
    x = torch.randn(50000, dtype=torch.float64, device="cuda")
    ref = x.clone()
    eng.allreduce_sum_(x)
    assert torch.equal(x, ref)
    # a sweep on a handle with a (one-rank) communicator equals the plain sweep bit for bit
    A, B, _ = fo.cfg3_problem(10, 8, 6)
    Q = fk.seeded_subspace(A.shape[0], 16)
    outs = []
    for e in (eng, engine):
        e.set_problem(A, B)
        Z, W = fk.feast_contour(0.0, 0.6, fk.feastdefault(fk.feastinit()))
        e.set_contour(Z, W, 2.0)
        e.set_real_projection(True)
        e.set_node_range(0, len(Z))
        e.set_solver("bicgstab", rtol=1e-12, atol=0.0, maxit=2000)
        dP, status, _ = e.contour_apply(e.upload(Q), 16)
        assert int(np.max(status)) == 0
        outs.append(e.download(dP, 16))
    assert np.array_equal(outs[0], outs[1])
    eng.comm_destroy()
    eng.close()


WORKER = r'''
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(240, exit=True)      # a wedged collective must not hang the suite
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch
import feast_oracle as fo, feastkit_jl_amd as fk
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), r"{out}"
eng = fk.HipEngine(0)
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    uid = eng.comm_unique_id()
    open(uidf + ".tmp", "wb").write(uid); os.rename(uidf + ".tmp", uidf)     # the host's own out-of-band channel
else:
    t0 = time.time()
    while not os.path.exists(uidf):
        time.sleep(0.01); assert time.time() - t0 < 60
    uid = open(uidf, "rb").read()
eng.comm_init(world, rank, uid, "shm")
assert (eng.comm_size, eng.comm_rank, eng.comm_transport()) == (world, rank, 2)
# plain reduce: values, growth of the staging buffer, rank-order determinism
x = torch.full((1000003,), float(rank + 1), dtype=torch.float64, device="cuda")
eng.allreduce_sum_(x)
assert bool((x == world * (world + 1) / 2).all())
g = torch.Generator(); g.manual_seed(100 + rank)
y = torch.randn(3 * 1000003, generator=g, dtype=torch.float64).cuda()      # larger than the first staging buffer
eng.allreduce_sum_(y)
ref = sum(torch.randn(3 * 1000003, generator=torch.Generator().manual_seed(100 + r), dtype=torch.float64) for r in range(world))
assert torch.allclose(y.cpu(), ref, rtol=0, atol=1e-12)
np.save(os.path.join(out, "y%d.npy" % rank), y.cpu().numpy())
zc = torch.full((4097,), complex(rank, -rank), dtype=torch.complex128, device="cuda")
eng.allreduce_sum_(zc)
assert bool((zc == complex(sum(range(world)), -sum(range(world)))).all())
assert eng.max_over_ranks(10.0 + rank) == 10.0 + world - 1
# the sweep: nodes split over the ranks, ONE packed reduce inside contour_apply, global status
A, B, _ = fo.cfg3_problem(10, 8, 6)
m = 24
Q = fk.seeded_subspace(A.shape[0], m)
Z, W = fk.feast_contour(0.0, 0.6, fk.feastdefault(fk.feastinit()))
def sweep(e, first, count, block=None, moments=False):
    e.set_problem(A, B); e.set_contour(Z, W, 2.0); e.set_real_projection(True)
    e.set_node_range(first, count); e.set_solver("bicgstab", rtol=1e-12, atol=0.0, maxit=3000)
    if block is not None: e.set_column_block(*block)
    r = e.contour_apply(e.upload(Q), m, None, want_moments=moments)
    e.set_column_block(0, -1)
    return r
first, count = fk.distribute_contour_points(len(Z), world)[rank]
dP, status, st = sweep(eng, first, count)
assert len(status) >= len(Z) and int(np.max(status)) == 0
solo = fk.HipEngine(0)
dR, sR, _ = sweep(solo, 0, len(Z))
P, R = eng.download(dP, m), solo.download(dR, m)
assert np.abs(P - R).max() <= 1e-11 * np.abs(R).max(), np.abs(P - R).max()
np.save(os.path.join(out, "p%d.npy" % rank), P)
# moments ride in the same reduce
dP2, status2, st2, Aq, Sq = sweep(eng, first, count, moments=True)
dR2, sR2, _, Aq0, Sq0 = sweep(solo, 0, len(Z), moments=True)
assert np.abs(Aq - Aq0).max() <= 1e-11 * np.abs(Aq0).max() and np.abs(Sq - Sq0).max() <= 1e-11 * np.abs(Sq0).max()
# column blocks: every rank sweeps ALL nodes for its own columns, the reduce fills in the rest
per = m // world
c0 = rank * per; cnt = (m - c0) if rank == world - 1 else per
dP3, status3, _ = sweep(eng, 0, len(Z), block=(c0, cnt))
# (narrower panels take a different reduction order inside the Krylov dots: the solves agree to rtol x cond, not bitwise)
assert np.abs(eng.download(dP3, m) - R).max() <= 1e-8 * np.abs(R).max()
solo.set_column_block(c0, cnt)
dR3, _, _ = sweep(solo, 0, len(Z), block=(c0, cnt))
assert np.array_equal(eng.download(dP3, m)[:, c0:c0 + cnt], solo.download(dR3, m)[:, c0:c0 + cnt])   # own block: bitwise the local sweep
# a node that does not converge on ONE rank is seen by every rank (no MAX reduce, no hang)
eng.set_problem(A, B); eng.set_contour(Z, W, 2.0); eng.set_node_range(first, count)
eng.set_solver("bicgstab", rtol=1e-10, atol=0.0, maxit=(2 if rank == world - 1 else 3000))
dP4, status4, _ = eng.contour_apply(eng.upload(Q), m)
lastfirst, lastcount = fk.distribute_contour_points(len(Z), world)[world - 1]
assert all(int(status4[e]) == 5 for e in range(lastfirst, lastfirst + lastcount)), status4
assert all(int(status4[e]) == 0 for e in range(0, lastfirst)), status4
eng.barrier()
eng.comm_destroy(); eng.close(); solo.close()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 3])
def test_shared_device_ranks_through_ctypes(engine, tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FEASTHIP_COMM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    ys = [np.load(tmp_path / f"y{r}.npy") for r in range(world)]
    ps = [np.load(tmp_path / f"p{r}.npy") for r in range(world)]
    assert all(np.array_equal(ys[0], y) for y in ys[1:])      # summed in rank order: bitwise identical on every rank
    assert all(np.array_equal(ps[0], p) for p in ps[1:])
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("feasthip_")]      # rendezvous segment unlinked


CFG3 = r'''
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(600, exit=True)
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch
import feastkit_jl_amd as fk
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), r"{out}"
eng = fk.HipEngine(0)
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    open(uidf + ".tmp", "wb").write(eng.comm_unique_id()); os.rename(uidf + ".tmp", uidf)
else:
    t0 = time.time()
    while not os.path.exists(uidf):
        time.sleep(0.01); assert time.time() - t0 < 120
eng.comm_init(world, rank, open(uidf, "rb").read(), "shm")
A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25, 0.1)
inside = lam[(lam >= 0.0) & (lam <= 0.1775)]
eng.set_problem(A, B)
fpm = fk.feastinit(); fpm[2], fpm[4], fpm[18] = 16, 40, {aspect}
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.1775, 64, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2, solver_maxiter={cap},
                           preloaded=True, node_assignment="balanced", column_groups="auto", real_projection=True)
res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
np.save(os.path.join(out, "c%d.npy" % rank), np.array([r.info, r.M, r.loop, r.epsout, res.max()] + list(np.sort(r.lambda_))))
assert r.info == 0 and r.M == len(inside) and np.abs(np.sort(r.lambda_) - inside).max() < 1e-10 and res.max() < 1e-10
eng.barrier(); eng.comm_destroy(); eng.close()
'''


@pytest.mark.parametrize("world,aspect,cap", [(2, 4000, 50), (4, 4000, 50), (4, 100, 100)])
def test_cfg3_full_size_ranks_on_one_card(tmp_path, world, aspect, cap):
    """BASELINE cfg 4's sharding at FULL size (N = 50 000, 16 nodes, M0 = 64) with 2 and 4 ranks on the one card of the
    test box, the bench's solver settings and layouts (2 -> 1x2, 4 -> 1x4 column groups): every rank returns all 44
    eigenpairs, eigenvalues within 1e-10 of the closed form, host-recomputed residual below 1e-10."""
    script = tmp_path / "cfg3.py"
    script.write_text(CFG3.format(root=ROOT, out=str(tmp_path), aspect=aspect, cap=cap))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FEASTHIP_COMM_TIMEOUT_S="300")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [np.load(tmp_path / f"c{r}.npy") for r in range(world)]
    assert all(np.array_equal(res[0], x) for x in res[1:])
    assert int(res[0][1]) == 44


REBAL = r'''
import os, sys, time, json, faulthandler
faulthandler.dump_traceback_later(300, exit=True)
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np
import feast_oracle as fo, feastkit_jl_amd as fk
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), r"{out}"
eng = fk.HipEngine(0)
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    open(uidf + ".tmp", "wb").write(eng.comm_unique_id()); os.rename(uidf + ".tmp", uidf)
else:
    t0 = time.time()
    while not os.path.exists(uidf):
        time.sleep(0.01); assert time.time() - t0 < 120
eng.comm_init(world, rank, open(uidf, "rb").read(), "shm")
A, B, lam = fo.cfg3_problem(16, 12, 10)
fpm = fk.feastinit(); fpm[2], fpm[4], fpm[18] = 16, 40, 1500
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.42, 48, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2, solver_maxiter=60,
                           node_assignment="balanced", column_groups=1, real_projection=True)
json.dump({{"info": r.info, "M": r.M, "eps": r.epsout, "lam": list(np.sort(r.lambda_)), "lists": r.stats["node_lists"],
           "its": r.stats["node_iterations"]}}, open(os.path.join(out, "b%d.json" % rank), "w"))
eng.barrier(); eng.comm_destroy(); eng.close()
'''


def test_node_groups_rebalance_between_loops(engine, tmp_path):
    """Two node groups on one card: after loop 1 the groups are rebuilt from the iteration counts that travelled in the
    packed reduce (identical on both ranks).  Every loop's two lists partition the 16 nodes, the lists do change, and the
    result is the single-rank result."""
    import json
    script = tmp_path / "rebal.py"
    script.write_text(REBAL.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FEASTHIP_COMM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    b = [json.load(open(tmp_path / f"b{r}.json")) for r in range(2)]
    A, B, lam = fo.cfg3_problem(16, 12, 10)
    inside = lam[(lam >= 0) & (lam <= 0.42)]
    for x in b:
        assert x["info"] == 0 and x["M"] == len(inside) and x["eps"] <= 1e-12
        assert np.allclose(x["lam"], inside, atol=1e-10)
    assert b[0]["lam"] == b[1]["lam"]
    assert len(b[0]["lists"]) == len(b[1]["lists"]) >= 3
    for l0, l1 in zip(b[0]["lists"], b[1]["lists"]):
        assert sorted(l0 + l1) == list(range(16))
    assert b[0]["lists"][0] == fk.balanced_contour_points(16, 2)[0]            # loops 0 and 1: the a-priori snake
    assert any(l != b[0]["lists"][0] for l in b[0]["lists"][2:])                # later loops: measured costs
    # the re-balanced groups carry comparable work
    last0, last1 = sum(b[0]["its"][-1]), sum(b[1]["its"][-1])
    assert abs(last0 - last1) <= 0.35 * (last0 + last1)


SPLIT = r'''
import os, sys, time, json, faulthandler
faulthandler.dump_traceback_later(300, exit=True)
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np
import feast_oracle as fo, feastkit_jl_amd as fk
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), r"{out}"
eng = fk.HipEngine(0)
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    open(uidf + ".tmp", "wb").write(eng.comm_unique_id()); os.rename(uidf + ".tmp", uidf)
else:
    t0 = time.time()
    while not os.path.exists(uidf):
        time.sleep(0.01); assert time.time() - t0 < 120
eng.comm_init(world, rank, open(uidf, "rb").read(), "shm")
A, B, lam = fo.cfg3_problem(16, 12, 10)
def layout(costs, world):          # the heaviest node's columns over ranks 0 and 1, every other node on rank 2
    heavy = [int(np.argmax(costs))]
    return [(heavy, 0, 2), (heavy, 1, 2), ([e for e in range(len(costs)) if e not in heavy], 0, 1)]
fpm = fk.feastinit(); fpm[2], fpm[4], fpm[18] = 16, 40, 1500
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.42, 48, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2, solver_maxiter=60,
                           node_assignment=layout, column_groups=1, real_projection=True)
json.dump({{"info": r.info, "M": r.M, "eps": r.epsout, "lam": list(np.sort(r.lambda_)), "lists": r.stats["node_lists"],
           "layout": r.stats.get("layout")}}, open(os.path.join(out, "s%d.json" % rank), "w"))
eng.barrier(); eng.comm_destroy(); eng.close()
'''


def test_heaviest_node_split_by_columns_over_two_ranks(engine, tmp_path):
    """Three ranks on one card with a layout callable: the slowest contour node's right-hand-side columns go to ranks 0
    and 1 (blocks [0, 32) and [32, 48)), the other 15 nodes to rank 2.  Every (node, column) pair is swept exactly once,
    so the packed reduce gives the single-rank Q_proj and the converged eigenvalues are the closed-form ones."""
    import json
    script = tmp_path / "split.py"
    script.write_text(SPLIT.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FEASTHIP_COMM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(3)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    b = [json.load(open(tmp_path / f"s{r}.json")) for r in range(3)]
    A, B, lam = fo.cfg3_problem(16, 12, 10)
    inside = lam[(lam >= 0) & (lam <= 0.42)]
    for x in b:
        assert x["info"] == 0 and x["M"] == len(inside) and x["eps"] <= 1e-12
        assert np.allclose(x["lam"], inside, atol=1e-10)
    assert b[0]["lam"] == b[1]["lam"] == b[2]["lam"]
    lay = b[0]["layout"]
    assert lay == b[1]["layout"] == b[2]["layout"] and lay[0][0] == lay[1][0] and len(lay[0][0]) == 1
    assert [lay[0][1:], lay[1][1:], lay[2][1:]] == [[0, 2], [1, 2], [0, 1]]
    assert sorted(lay[0][0] + lay[2][0]) == list(range(16))
    assert b[0]["lists"][-1] == b[1]["lists"][-1] == lay[0][0] and b[2]["lists"][-1] == lay[2][0]


CFG5 = r'''
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(300, exit=True)
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np
import feastkit_jl_amd as fk
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), r"{out}"
eng = fk.HipEngine(0)
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    open(uidf + ".tmp", "wb").write(eng.comm_unique_id()); os.rename(uidf + ".tmp", uidf)
else:
    t0 = time.time()
    while not os.path.exists(uidf):
        time.sleep(0.01); assert time.time() - t0 < 120
eng.comm_init(world, rank, open(uidf, "rb").read(), "shm")
A, delta = fk.workloads.disc_spectrum_general(512)
inside = delta[np.abs(delta) <= 2.0]
key = lambda x: (round(x.real, 7), round(x.imag, 7))
res = []
for prec in (64, 32):
    fpm = fk.feastinit(); fpm[8] = 24; fpm[4] = 20
    r = fk.feast_hip_general(eng, A, None, 0.0, 2.0, 48, fpm, inner_precision=prec)
    assert r.info == 0 and r.M == len(inside), (r.info, r.M, len(inside))
    err = np.abs(np.array(sorted(r.lambda_, key=key)) - np.array(sorted(inside, key=key))).max()
    hres = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert err < 1e-10 and hres.max() < 1e-10, (err, hres.max())
    res += [r.M, r.loop] + [v for x in sorted(r.lambda_, key=key) for v in (x.real, x.imag)]
np.save(os.path.join(out, "g%d.npy" % rank), np.array(res))
eng.barrier(); eng.comm_destroy(); eng.close()
'''


def test_cfg5_split_four_ranks_general_lu(tmp_path):
    """BASELINE cfg 5's split -- 24 full-contour nodes over 4 ranks, 6 per rank, batched complex LU (complex128 and
    complex64 + fp64 refinement), M0 = 48 -- at N = 512 on one card: every rank returns the eigenvalues inside the
    circle (known by construction) to 1e-10, identical on all ranks."""
    script = tmp_path / "cfg5.py"
    script.write_text(CFG5.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FEASTHIP_COMM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(4)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    g = [np.load(tmp_path / f"g{r}.npy") for r in range(4)]
    assert all(np.array_equal(g[0], x) for x in g[1:])


CFG4 = r'''
import os, sys, time, threading, faulthandler
faulthandler.dump_traceback_later(800, exit=True)
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch
import feastkit_jl_amd as fk
proc, nproc, tpp, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), r"{out}"
world = nproc * tpp
A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25, 0.1)
inside = lam[(lam >= 0.0) & (lam <= 0.1775)]
uidf = os.path.join(out, "uid.bin")
results, errors = {{}}, []

def run(t):
    rank = proc * tpp + t
    try:
        eng = fk.HipEngine(0)
        if rank == 0:
            open(uidf + ".tmp", "wb").write(eng.comm_unique_id()); os.rename(uidf + ".tmp", uidf)
        else:
            t0 = time.time()
            while not os.path.exists(uidf):
                time.sleep(0.01); assert time.time() - t0 < 120
        eng.comm_init(world, rank, open(uidf, "rb").read(), "shm")
        eng.set_problem(A, B)
        fpm = fk.feastinit(); fpm[2], fpm[4], fpm[18] = 16, 40, {aspect}
        r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.1775, 64, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2, solver_maxiter={cap},
                                   preloaded=True, node_assignment="balanced", column_groups="auto", real_projection=True)
        res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
        results[rank] = np.array([r.info, r.M, r.loop, r.epsout, res.max()] + list(np.sort(r.lambda_)))
        assert r.info == 0 and r.M == len(inside) and np.abs(np.sort(r.lambda_) - inside).max() < 1e-10 and res.max() < 1e-10
        eng.barrier(); eng.comm_destroy(); eng.close()
    except BaseException as exc:                    # a failed rank must not leave its peers waiting in the collective for ever
        errors.append((rank, repr(exc)))
        os._exit(3)

threads = [threading.Thread(target=run, args=(t,)) for t in range(tpp)]
for th in threads: th.start()
for th in threads: th.join()
assert not errors, errors
for rank, v in results.items():
    np.save(os.path.join(out, "c%d.npy" % rank), v)
'''


@pytest.mark.parametrize("nproc,tpp", [(1, 8), (4, 2)])
def test_cfg4_eight_ranks_on_one_card(tmp_path, nproc, tpp):
    """BASELINE cfg 4 itself -- the 50 000-unknown problem, 16 quadrature nodes sharded over EIGHT ranks, the layout bench.py
    takes at N = 8 (2 node groups x 4 column groups of 16 right-hand sides, the bench's solver settings) -- on the one card
    of the test box.  The box admits at most six GPU processes, so the eight ranks are eight handles on eight host threads of
    one process, or two threads in each of four processes (the shared-device transport reaches a same-process peer through
    its pointer, another process's through hipIpc).  Every rank returns all 44 eigenpairs, bitwise the same on every rank,
    eigenvalues within 1e-10 of the closed form, host-recomputed residual below 1e-10."""
    script = tmp_path / "cfg4.py"
    script.write_text(CFG4.format(root=ROOT, out=str(tmp_path), aspect=4000, cap=50))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FEASTHIP_COMM_TIMEOUT_S="300")
    procs = [subprocess.Popen([sys.executable, str(script), str(p), str(nproc), str(tpp)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for p in range(nproc)]
    outs = [p.communicate(timeout=1000)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [np.load(tmp_path / f"c{r}.npy") for r in range(8)]
    assert all(np.array_equal(res[0], x) for x in res[1:])
    assert int(res[0][0]) == 0 and int(res[0][1]) == 44


RCCL2 = r'''
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(300, exit=True)
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch
import feast_oracle as fo, feastkit_jl_amd as fk
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), r"{out}"
eng = fk.HipEngine(rank)                               # one rank per GPU
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    open(uidf + ".tmp", "wb").write(eng.comm_unique_id()); os.rename(uidf + ".tmp", uidf)
else:
    t0 = time.time()
    while not os.path.exists(uidf):
        time.sleep(0.01); assert time.time() - t0 < 120
eng.comm_init(world, rank, open(uidf, "rb").read(), "rccl")
assert (eng.comm_size, eng.comm_rank, eng.comm_transport()) == (world, rank, 1)
x = torch.full((1000003,), float(rank + 1), dtype=torch.float64, device=eng.device)
eng.allreduce_sum_(x)
assert bool((x == world * (world + 1) / 2).all())
A, B, lam = fo.cfg3_problem(16, 12, 10)
inside = lam[(lam >= 0) & (lam <= 0.42)]
fpm = fk.feastinit(); fpm[2], fpm[4], fpm[18] = 16, 40, 1500
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.42, 48, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2, solver_maxiter=60,
                           node_assignment="balanced", column_groups="auto", real_projection=True)
assert r.info == 0 and r.M == len(inside) and np.abs(np.sort(r.lambda_) - inside).max() < 1e-10 and r.epsout <= 1e-12
np.save(os.path.join(out, "r%d.npy" % rank), np.array([r.info, r.M, r.loop, r.epsout] + list(np.sort(r.lambda_))))
eng.barrier(); eng.comm_destroy(); eng.close()
'''


def test_rccl_two_ranks_two_gpus(tmp_path):
    """The RCCL transport with MORE than one rank (fh_comm.hip: ncclCommInitRank / ncclAllReduce over xGMI): needs two GPUs,
    one rank each -- skipped on the one-card test box, executed by the first multi-GPU box that runs the suite."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks of one communicator on one device)")
    script = tmp_path / "rccl2.py"
    script.write_text(RCCL2.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    a, b = (np.load(tmp_path / f"r{r}.npy") for r in range(2))
    assert np.array_equal(a, b)


CPLX2 = r'''
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(300, exit=True)
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, scipy.sparse as sp
import feastkit_jl_amd as fk
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), r"{out}"
eng = fk.HipEngine(0)
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    open(uidf + ".tmp", "wb").write(eng.comm_unique_id()); os.rename(uidf + ".tmp", uidf)
else:
    t0 = time.time()
    while not os.path.exists(uidf):
        time.sleep(0.01); assert time.time() - t0 < 120
eng.comm_init(world, rank, open(uidf, "rb").read(), "shm")
# complex Hermitian pencil, complex subspace: the projection keeps its imaginary part
n, m = 1500, 40
rng = np.random.default_rng(5)
o1 = 0.3 * (rng.standard_normal(n - 1) + 1j * rng.standard_normal(n - 1))
o2 = 0.1 * (rng.standard_normal(n - 7) + 1j * rng.standard_normal(n - 7))
A = sp.csr_matrix(sp.diags([o2.conj(), o1.conj(), np.linspace(0.0, 40.0, n), o1, o2], [-7, -1, 0, 1, 7]))
B = sp.csr_matrix(sp.diags([0.05 * np.ones(n - 1), 1.0 + 0.2 * rng.random(n), 0.05 * np.ones(n - 1)], [-1, 0, 1]).astype(np.complex128))
Q = fk.seeded_subspace(n, m, complex_values=True)
fpm = fk.feastinit(); fpm[2] = 8; fk.feastdefault(fpm)
Z, W = fk.feast_contour(3.0, 4.0, fpm)
solo = fk.HipEngine(0)
for e in (eng, solo):
    e.set_problem(A, B); e.set_contour(Z, W, 2.0); e.set_solver("bicgstab", rtol=1e-13, atol=0.0, maxit=6000)
for realproj in (False, True):
    for layout in ("columns", "nodes"):
        for e in (eng, solo):
            e.set_real_projection(realproj)
        solo.set_node_range(0, len(Z))
        dR, _, _ = solo.contour_apply(solo.upload(Q), m)
        R = solo.download(dR, m)
        if layout == "columns":                     # every rank sweeps all nodes for its block of columns (16 + 24)
            eng.set_node_range(0, len(Z))
            eng.set_column_block(0 if rank == 0 else 16, 16 if rank == 0 else m - 16)
        else:                                       # every rank sweeps its nodes for all columns
            first, count = fk.distribute_contour_points(len(Z), world)[rank]
            eng.set_node_range(first, count)
        status, _ = eng.contour_apply_resident(eng.upload(Q), m)
        eng.set_column_block(0, -1)
        assert int(np.max(status)) == 0
        P = eng.download(eng.export_resident(m, which=1), m)
        assert np.abs(P - R).max() <= 1e-9 * np.abs(R).max(), (realproj, layout, np.abs(P - R).max())
        if not realproj:
            assert np.abs(P.imag).max() > 1e-3 * np.abs(P).max()       # really a complex panel
        np.save(os.path.join(out, "h%d_%d_%s.npy" % (rank, int(realproj), layout)), P)
        # the reduction + Ritz step on the summed panel agree with the single-rank per-primitive path
        rk, Sq, Aq = eng.rr_reduce_resident(m, 1.5e-8)
        dRo = dR.clone()
        assert solo.orthonormalize(dRo, m, 1.5e-8) == rk
        S2, A2 = solo.project(dRo, rk)
        import scipy.linalg as sla
        assert np.abs(sla.eigh(Sq, Aq, eigvals_only=True) - sla.eigh(S2, A2, eigvals_only=True)).max() <= 1e-8
eng.barrier(); eng.comm_destroy(); eng.close(); solo.close()
'''


def test_two_ranks_complex_projection_resident_sweep(tmp_path):
    """The resident sweep under a communicator with a COMPLEX projection (complex Hermitian pencil, complex subspace: the packed
    reduce carries interleaved complex panels) and with the real projection, by column blocks (16 + 24 of 40 columns: two
    panel widths) and by node blocks: the summed resident Q_proj equals the single-rank sweep on both ranks, bitwise the same
    on both, and the resident reduction on it gives the per-primitive path's Ritz values."""
    script = tmp_path / "cplx2.py"
    script.write_text(CPLX2.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FEASTHIP_COMM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for rp in (0, 1):
        for layout in ("columns", "nodes"):
            a, b = (np.load(tmp_path / f"h{r}_{rp}_{layout}.npy") for r in range(2))
            assert np.array_equal(a, b)
