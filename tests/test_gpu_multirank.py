"""Two ranks sharing the one GPU of the test box (gloo rendezvous, CUDA tensors): the real
HipEngine path of the sharded sweep -- node lists per rank, all-reduce of Q_proj, redundant
reduced eigenproblem -- must reproduce the single-rank result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import feast_oracle as fo
import feastkit_jl_amd as fk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch, torch.distributed as dist
import feast_oracle as fo, feastkit_jl_amd as fk
rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
A, B, lam = fo.cfg3_problem(16, 12, 10)
inside = lam[(lam >= 0) & (lam <= 0.42)]
out = []
for assign, cg in (("block", 1), ("balanced", 1), ("balanced", 2)):
    eng = fk.HipEngine(0)
    fpm = fk.feastinit(); fpm[2] = 8; fpm[4] = 40
    r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.42, 32, fpm, solver="bicgstab", warm_start=True,
                               inner_rtol=1e-2, solver_maxiter=100, node_assignment=assign, column_groups=cg)
    out += [r.info, r.M, r.epsout] + list(np.sort(r.lambda_))
    eng.close()
# dense LU path, general problem (full contour, 2 ranks x 8 nodes)
Ag = np.diag([0.5+0.1j, 1.0+0.2j, 2.0-0.1j, 4.0]) + 0.01 * np.triu(np.ones((4, 4)), 1)
eng = fk.HipEngine(0)
g = fk.feast_hip_general(eng, Ag, None, 1.0+0.1j, 1.3, 4, fk.feastinit())
out += [g.info, g.M] + list(np.sort(g.lambda_.real))
# sparse direct solver (blocked band LU after reverse Cuthill-McKee), 8 nodes over 2 ranks: each rank factors its own 4
os.environ["FH_WBAND"] = "1"
eng = fk.HipEngine(0)
fpm = fk.feastinit(); fpm[2] = 8
d = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.42, 32, fpm, solver="banded")
out += [d.info, d.M, d.epsout, d.stats.get("factorizations", -1)] + list(np.sort(d.lambda_))
eng.close()
# the same through the multifrontal plan of the sparse direct solver
del os.environ["FH_WBAND"]
os.environ["FH_MF"] = "1"
eng = fk.HipEngine(0)
d = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.42, 32, fpm, solver="banded")
out += [d.info, d.M, d.epsout, d.stats.get("factorizations", -1), eng.band_plan()[3]] + list(np.sort(d.lambda_))
eng.close()
np.save(r"{out}/g%d.npy" % rank, np.array(out, dtype=float))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_ranks_one_gpu_match_single_rank(engine, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)
    A, B, lam = fo.cfg3_problem(16, 12, 10)
    inside = lam[(lam >= 0) & (lam <= 0.42)]
    fpm = fk.feastinit(); fpm[2] = 8; fpm[4] = 40
    one = fk.feast_hip_hermitian(engine, A, B, 0.0, 0.42, 32, fpm, solver="bicgstab", warm_start=True,
                                 inner_rtol=1e-2, solver_maxiter=100)
    n = len(inside)
    for off in (0, 3 + n, 2 * (3 + n)):
        assert (int(g0[off]), int(g0[off + 1])) == (0, n) == (one.info, one.M)
        assert g0[off + 2] <= 1e-12
        assert np.allclose(g0[off + 3: off + 3 + n], inside, atol=1e-10)
        assert np.allclose(g0[off + 3: off + 3 + n], np.sort(one.lambda_), atol=1e-10)
    off = 3 * (3 + n)
    assert (int(g0[off]), int(g0[off + 1])) == (0, 3) and np.allclose(g0[off + 2: off + 5], [0.5, 1.0, 2.0], atol=1e-8)
    off += 5
    assert (int(g0[off]), int(g0[off + 1])) == (0, n) and g0[off + 2] <= 1e-12 and int(g0[off + 3]) == 4     # 4 of the 8 nodes factored per rank
    assert np.allclose(g0[off + 4: off + 4 + n], inside, atol=1e-10)
    off += 4 + n                                                          # multifrontal plan: same result, 4 factorisations per rank
    assert (int(g0[off]), int(g0[off + 1])) == (0, n) and g0[off + 2] <= 1e-12 and int(g0[off + 3]) == 4 and int(g0[off + 4]) == 2
    assert np.allclose(g0[off + 5: off + 5 + n], inside, atol=1e-10)


WORKER4 = r'''
import os, sys
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, torch, torch.distributed as dist
import feast_oracle as fo, feastkit_jl_amd as fk
rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size={world})
A, B, lam = fo.cfg3_problem(16, 12, 10)
eng = fk.HipEngine(0)
fpm = fk.feastinit(); fpm[2] = 16; fpm[4] = 40
# the bench's settings: COCG in sum mode, Ritz warm start, inexact solves, balanced nodes, column groups by rule
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.42, 64, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2,
                           solver_maxiter=100, node_assignment="balanced", column_groups="auto", real_projection=True)
np.save(r"{out}/w%d.npy" % rank, np.array([r.info, r.M, r.epsout, r.loop] + list(np.sort(r.lambda_)), dtype=float))
eng.close()
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [3, 4])
def test_bench_layout_ranks_match_single_rank(engine, tmp_path, world):
    """The layouts bench.py takes at N = 4 (1 node group x 4 column groups of 16 columns) and at an odd rank count
    (3 node groups, no column split), with the bench's solver settings, on one card: every rank returns the
    single-rank eigenvalues."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker4.py"
    script.write_text(WORKER4.format(root=ROOT, port=port, out=str(tmp_path), world=world))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [np.load(tmp_path / f"w{r}.npy") for r in range(world)]
    assert all(np.array_equal(res[0], x) for x in res[1:])
    A, B, lam = fo.cfg3_problem(16, 12, 10)
    inside = lam[(lam >= 0) & (lam <= 0.42)]
    fpm = fk.feastinit(); fpm[2] = 16; fpm[4] = 40
    one = fk.feast_hip_hermitian(engine, A, B, 0.0, 0.42, 64, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2,
                                 solver_maxiter=100, real_projection=True)
    n = len(inside)
    assert (int(res[0][0]), int(res[0][1])) == (0, n) == (one.info, one.M)
    assert res[0][2] <= 1e-12 and abs(int(res[0][3]) - one.loop) <= 2
    assert np.allclose(res[0][4:4 + n], inside, atol=1e-10)
    assert np.allclose(res[0][4:4 + n], np.sort(one.lambda_), atol=1e-10)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_command_two_ranks_on_one_card(launcher):
    """The command the driver's scaling run issues, at N = 2 on the one card of the test box (ranks share the device: the
    library takes its shared-device transport): `python bench.py --gpus 2 ...` launching its own ranks, and the driver's
    own form `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...`.  The JSON line must say
    n_gpus == 2 and carry the full answer (44 / 44 eigenpairs, converged, residual <= 1e-10)."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--headline-only"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, text=True)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "strong"
    assert d["eigenpairs"] == d["expected_eigenpairs"] == 44 and d["converged"] is True
    assert d["max_residual"] <= 1e-10 and d["max_eigenvalue_error"] <= 1e-10
    assert d["value"] > 0 and d["unit"] == "eigenpairs/s"
