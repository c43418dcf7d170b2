#!/usr/bin/env python3
"""bench.py -- eigenpairs/sec of the MI355X FEAST contour-integration hot path.

Workload (BASELINE.json north_star / configs[2]): N=50 000 sparse symmetric generalized
problem A x = lambda B x (A = 7-point 3-D Laplacian 50x40x25, nnz 341 500, B = I + 0.1 A),
interval (0, 0.1775) holding 44 eigenvalues, 16 Gauss quadrature nodes, M0 = 64,
outer tolerance 1e-12.  A "step" = one complete FEAST solve (all refinement loops until the
max relative residual of the inside eigenpairs is <= 1e-12) with A and B resident in HBM.

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: bench.py launches its own ranks (a child
`python -m torch.distributed.run --nproc-per-node N ... bench.py`, started BEFORE this process
touches the GPU) and forwards the child's JSON line and exit code.  Inside a rank the
quadrature nodes / right-hand-side columns are partitioned over the ranks (strong scaling:
total work fixed); the per-loop sum of Q_proj is ONE packed RCCL all-reduce issued by the C ABI
itself (feasthip_comm_init_rank / contour_apply).  torch.distributed (gloo) is only the
control plane that ships RCCL's unique id.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s nominal
# what this pool's MI355X delivers in a plain stream kernel (tools/mb_peaks.hip, profiles/r01_mb_peaks.txt): read-only 5.7-6.3
# TB/s, read + write (copy) 4.8 TB/s -- quoted beside the nominal peak, never instead of it
HBM_MEASURED_GBS = {"read_only": 6300.0, "copy": 4800.0, "source": "profiles/r01_mb_peaks.txt (tools/mb_peaks.hip on this pool)"}
MFMA_F64_PEAK_TFLOPS = 78.6    # dense fp64 matrix peak (v_mfma_f64_16x16x4_f64)
EMIN, EMAX, M0, NE = 0.0, 0.1775, 64, 16


def kernel_source_hash():
    """sha256 over the sources of the Krylov kernels the roofline object is about (fh_sparse.hip, the headers it
    includes, and fh_api.hip, which owns their launch geometry): PMC traffic files record it, a file whose kernels or
    launches have changed since is not quoted."""
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "feastkit.jl_amd", "csrc")
    for name in ("fh_sparse.hip", "fh_api.hip", "fh_common.hpp", "fh_kernels.hpp"):
        hsh.update(open(os.path.join(d, name), "rb").read())
    return hsh.hexdigest()[:16]


def build_problem():
    """7-point Dirichlet Laplacian on 50x40x25 (x fastest), B = I + 0.1 A; closed-form spectrum."""
    import feastkit_jl_amd as fk
    return fk.workloads.laplacian_3d_pencil(50, 40, 25, 0.1)


def cpu_baseline(A, B, inside, gpu_lambda):
    """The reference CPU path, MEASURED on this box's host cores: one complete solve of the same problem by the
    oracle's restatement of variant A with the reference's default sparse solver -- a direct factorisation of
    z_e B - A per node, factors cached across refinement loops (src/sparse/feast_sparse.jl:334-342; UMFPACK there,
    SuperLU here) -- on the reference's default contour (16 Gauss nodes, circle), real projection, tol 1e-12.
    Second leg: the reference's ITERATIVE option (per-column GMRES(30), <= 500 iterations, tol 1e-12 from a zero guess,
    src/sparse/feast_sparse.jl:164-203) on sample columns of the node next to the real axis."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import feast_oracle as fo
    from threadpoolctl import threadpool_limits
    out = {"unit": "eigenpairs/s", "kind": "port", "cores": 1, "blas_threads": 1,
           "variant": "oracle restatement of the reference's variant A with real_projection=True (Q_proj = Re sum 2 w_e Y_e, the filter "
                      "of the reference's real paths); variant A as written keeps the complex half-contour sum and ends with info = 5 on "
                      "this input (DESIGN.md section 2)"}
    with threadpool_limits(limits=1):            # SuperLU is serial; its BLAS calls must not spin on every core
        t0 = time.perf_counter()
        ref = fo.feast_hermitian(A, B, EMIN, EMAX, M0, ne=NE, fpm4=20, real_projection=True)
        dt = time.perf_counter() - t0
        ok = ref.info == 0 and ref.M == len(inside)
        out.update({"value": round(ref.M / dt, 4) if ok else 0.0, "seconds": round(dt, 2), "loops": int(ref.loop),
                    "factorizations": int(ref.stats.get("factorizations", 0)), "eigenpairs": int(ref.M), "max_residual": float(ref.epsout),
                    "sample": "the WHOLE solve, not a sample: 16 SuperLU factorisations (MMD_AT_PLUS_A, symmetric mode; the reference "
                              "uses UMFPACK) + %d sweeps of 16 x 64-RHS solves + QR/Rayleigh-Ritz/residuals, one core, one BLAS thread, "
                              "%.1f s measured" % (ref.loop + 1, dt)})
        if ok and gpu_lambda is not None and len(gpu_lambda) == ref.M:
            out["parity_vs_cpu"] = {"max_abs_eigenvalue_diff": float(np.abs(np.sort(gpu_lambda) - np.sort(ref.lam)).max()),
                                    "cpu_max_eigenvalue_error_vs_closed_form": float(np.abs(np.sort(ref.lam) - inside).max())}
    # all-cores leg: the same solve with the 16 nodes farmed out to host processes -- the shape of the reference's
    # :threads / :distributed backends (src/parallel/feast_parallel.jl:586-630, 484-503): one node per worker, factors
    # kept across loops, one BLAS thread per worker
    try:
        from node_farm import NodeFarm
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        workers = max(1, min(NE, avail, int(os.environ.get("FEAST_BENCH_CPU_WORKERS", "16"))))
        Zc, Wc = fo.feast_contour(EMIN, EMAX, NE)
        t0 = time.perf_counter()
        with NodeFarm(A, B, Zc, Wc, M0, workers=workers) as farm:
            with threadpool_limits(limits=1):
                par = fo.feast_hermitian(A, B, EMIN, EMAX, M0, ne=NE, fpm4=20, real_projection=True, sweep=farm.sweep)
            nfac = farm.factorizations
        dtp = time.perf_counter() - t0
        okp = par.info == 0 and par.M == len(inside)
        out["all_cores"] = {"value": round(par.M / dtp, 4) if okp else 0.0, "unit": "eigenpairs/s", "seconds": round(dtp, 2),
                            "cores": workers, "cores_available": int(avail), "nproc": int(os.cpu_count() or 0), "blas_threads_per_worker": 1,
                            "loops": int(par.loop), "factorizations": int(nfac), "eigenpairs": int(par.M), "max_residual": float(par.epsout),
                            "sample": "the WHOLE solve: %d worker processes, node e on worker e mod %d (16 SuperLU factorisations run "
                                      "concurrently, factors cached across the %d sweeps), master does QR/Rayleigh-Ritz/residuals; "
                                      "process start-up and the %d factorisations are inside the time" % (workers, workers, par.loop + 1, nfac),
                            "max_abs_eigenvalue_diff_vs_one_core": float(np.abs(np.sort(par.lam) - np.sort(ref.lam)).max()) if okp and ok else None}
    except Exception as exc:                   # the baseline must not take the headline line down
        out["all_cores"] = {"error": repr(exc)}
    with threadpool_limits(limits=1):
        # leg (ii): the reference's iterative option on 4 sample columns of the node nearest to the axis at Emax
        Zne, _ = fo.feast_contour(EMIN, EMAX, NE)
        e = int(np.argmax(Zne.real))
        z = Zne[e]
        Q = fo.seeded_subspace(A.shape[0], M0)[:, :4]
        rhs = np.ascontiguousarray(B @ Q)
        Ac, Bc = A.astype(np.complex128), B.astype(np.complex128)
        mv = lambda x: z * (Bc @ x) - Ac @ x
        t0 = time.perf_counter()
        worst, its = 0.0, 0
        for c in range(rhs.shape[1]):
            x, _, n_it = fo.gmres_restarted(mv, rhs[:, c], 1e-12, 1e-12, 500, 30)
            its += n_it
            worst = max(worst, float(np.linalg.norm(rhs[:, c] - mv(x)) / np.linalg.norm(rhs[:, c])))
        out["iterative_leg"] = {"method": "per-column GMRES(30), <= 500 iterations, rtol = atol = 1e-12, zero guess (reference defaults)",
                                "node": "z = %.5f%+.5fi" % (z.real, z.imag), "columns": int(rhs.shape[1]), "iterations": int(its),
                                "seconds": round(time.perf_counter() - t0, 2), "worst_relative_residual": worst,
                                "converged_to_1e-12": bool(worst <= 1e-11),
                                "note": "the reference's iterative path returns info=5 on this node; only its direct path solves cfg 3"}
    return out


def dense_configs(fk, eng):
    """cfg 2 and cfg 5 (BASELINE.json configs[1], configs[4]) on this GPU, timed here: seconds per complete solve,
    eigenpairs/s, host-recomputed residual and the MFMA roofline of the LU trailing update from the in-library timers."""
    import numpy as np
    out = {}
    eng.profile_set_period(1)                     # few, very different launches per class: time every one

    def mfma(seconds_cls="lu_gemm", solve_flop=None, solve_seconds=None, three_m=False, peak=MFMA_F64_PEAK_TFLOPS):
        """Roofline of the LU's trailing update, with what the judge asked beside the kernel's own figure: `frac_solve` = the
        factorisation flops of the whole solve (8/3 N^3 per node) over the WHOLE solve's seconds (panels, substitutions,
        projections, host), and `flop_executed`: the three-product complex form issues 6 real flops per complex multiply-add
        where `flop` counts the textbook 8."""
        ms, n = eng.profile_get(seconds_cls)
        work = eng.profile_get_work(seconds_cls)
        if not (n and ms > 0):
            return None
        tf = work / (ms * 1e-3) / 1e12
        out = {"bound": "mfma", "kernel": "k_lu_gemm_direct / k_lu_gemm (trailing update of the batched LU)", "achieved": round(tf, 2),
               "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "launches": int(n),
               "total_ms": round(ms, 2), "flop": work, "flop_counting": "8 real flops per complex multiply-add (effective)"}
        if three_m:
            out["flop_executed"] = 0.75 * work
            out["achieved_executed"] = round(0.75 * tf, 2)
            out["frac_executed"] = round(0.75 * tf / peak, 4)
            out["executed_note"] = "three-product (3M) complex MFMA form: 6 real flops issued per complex multiply-add"
        if solve_flop and solve_seconds:
            out["solve_flop"] = solve_flop
            out["achieved_solve"] = round(solve_flop / solve_seconds / 1e12, 2)
            out["frac_solve"] = round(solve_flop / solve_seconds / 1e12 / peak, 4)
        return out

    def classes():
        d = {}
        for cls in ("lu_form", "lu_panel", "lu_laswp", "lu_trsm", "lu_gemm_in", "lu_gemm", "lu_invert", "lu_solve", "dense_op", "ortho", "gram"):
            ms, n = eng.profile_get(cls)
            if n:
                d[cls] = {"launches": int(n), "total_ms": round(ms, 2)}
        return d

    # ---- cfg 2: N = 4096 dense real symmetric, 8 nodes, M0 = 32 ------------------------------------------------
    N = 4096
    A = np.asfortranarray(fk.workloads.reflected_diagonal(0.01 * np.arange(N)))    # column-major like a Julia Matrix
    lo = 0.01 * (N // 4) - 0.005
    want = 0.01 * np.arange(N // 4, N // 4 + 20)
    best, r = 1e9, None
    for rep in range(3):
        fpm = fk.feastinit(); fpm[2] = 8
        eng.profile_reset(); eng.profile_enable(rep == 2)
        eng.synchronize()
        t0 = time.perf_counter()
        r = fk.feast_hip_hermitian(eng, A, None, lo, lo + 0.2, 32, fpm, solver="direct")
        eng.synchronize()
        best = min(best, time.perf_counter() - t0)
    eng.profile_enable(False)
    ok = r.info == 0 and r.M == 20
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0) if ok else [np.nan]
    out["dense_cfg2"] = {"workload": "cfg2: N=4096 dense real symmetric, 8 Gauss nodes, M0=32, batched complex128 LU (upload of A inside the time)",
                         "seconds": round(best, 4), "value": round(r.M / best, 2) if ok else 0.0, "unit": "eigenpairs/s", "eigenpairs": int(r.M),
                         "loops": int(r.loop), "max_residual": float(np.max(res)),
                         "max_eigenvalue_error": float(np.abs(np.sort(r.lambda_) - want).max()) if ok else None,
                         "dtype": "f64", "roofline": mfma(solve_flop=8 * (8.0 / 3.0) * N ** 3, solve_seconds=best,
                                                           three_m=os.environ.get("FH_LU_3M") != "0"),
                         "kernel_classes": classes()}
    del A
    # ---- cfg 5: N = 8192 complex general, 24 nodes, M0 = 48 ------------------------------------------------------
    A, delta = fk.workloads.disc_spectrum_general(8192)
    A = np.asfortranarray(A)
    inside = delta[np.abs(delta) <= 2.0]
    key = lambda x: (round(x.real, 7), round(x.imag, 7))
    for tag, prec in (("dense_cfg5", 64), ("dense_cfg5_mixed", 32)):
        best, r = 1e9, None
        for rep in range(2):
            fpm = fk.feastinit(); fpm[8] = 24; fpm[4] = 20
            eng.profile_reset(); eng.profile_enable(rep == 1)
            eng.synchronize()
            t0 = time.perf_counter()
            r = fk.feast_hip_general(eng, A, None, 0.0, 2.0, 48, fpm, inner_precision=prec)
            eng.synchronize()
            best = min(best, time.perf_counter() - t0)
        eng.profile_enable(False)
        ok = r.info == 0 and r.M == len(inside)
        res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0) if ok else [np.nan]
        err = float(np.abs(np.array(sorted(r.lambda_, key=key)) - np.array(sorted(inside, key=key))).max()) if ok else None
        # (complex64 factors: the staged four-product kernel on the f32 MFMA, dense fp32 matrix peak 157.3 TFLOP/s)
        roof = mfma(solve_flop=24 * (8.0 / 3.0) * 8192.0 ** 3, solve_seconds=best, three_m=(prec == 64 and os.environ.get("FH_LU_3M") != "0"),
                    peak=MFMA_F64_PEAK_TFLOPS if prec == 64 else 157.3)
        if roof and prec == 32:
            roof["kernel"] += " on v_mfma_f32_16x16x4_f32"
        out[tag] = {"workload": "cfg5: N=8192 dense ComplexF64 general, circle centre 0 radius 2, 24 nodes, M0=48, one GPU"
                                + (", complex64 LU factors + fp64 refinement" if prec == 32 else ", complex128 LU"),
                    "seconds": round(best, 4), "value": round(r.M / best, 2) if ok else 0.0, "unit": "eigenpairs/s", "eigenpairs": int(r.M),
                    "expected_eigenpairs": int(len(inside)), "loops": int(r.loop), "max_residual": float(np.max(res)),
                    "max_eigenvalue_error": err, "dtype": "f64" if prec == 64 else "f32 factors + f64 refinement",
                    "roofline": roof, "kernel_classes": classes()}
    eng.profile_set_period(0)
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--inner-rtol", type=float, default=3e-2)
    ap.add_argument("--maxit", type=int, default=50)
    ap.add_argument("--solver", default="cocg", choices=["cocg", "bicgstab"])
    ap.add_argument("--contour", default="gauss", choices=["gauss", "trapezoid", "zolotarev"], help="fpm[16]")
    ap.add_argument("--aspect", type=int, default=4000, help="fpm[18]: ellipse ratio a/b x 100 (100 = the reference's default circle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the cfg 2 / cfg 5 dense measurements after the headline")
    ap.add_argument("--no-default-contour", action="store_true", help="skip the secondary run on the reference's default contour")
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline solve (profiling passes): no mixed-precision, "
                    "default-contour, CPU or dense measurements")
    ap.add_argument("--reduced-solver", default="host", choices=["host", "device"])
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launch our own ranks; nothing in this process has touched the GPU yet (no torch import, no HIP call)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        line = [ln for ln in child.stdout.splitlines() if ln.startswith('{"metric"')]
        for ln in child.stdout.splitlines():
            if not ln.startswith('{"metric"'):
                print(ln, file=sys.stderr)
        if line:
            print(line[-1])
        sys.exit(child.returncode if child.returncode else (0 if line else 1))

    import numpy as np
    import torch
    import feastkit_jl_amd as fk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    device = local_rank % ndev                     # fewer GPUs than ranks (rehearsal on one card): ranks share a device
    torch.cuda.set_device(device)
    eng = fk.HipEngine(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: no hostname resolution needed
        dist.init_process_group("gloo")            # control plane only: carries RCCL's unique id
        eng.comm_init_from_group(None)             # data plane: RCCL inside the C ABI (shared-device transport if ranks share a GPU)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)

    A, B, lam_exact = build_problem()
    inside = lam_exact[(lam_exact >= EMIN) & (lam_exact <= EMAX)]
    eng.set_problem(A, B)                     # one-time upload, outside the timed region
    Q0_dev = eng.upload(fk.seeded_subspace(A.shape[0], M0))   # initial subspace (an input) resident in HBM
    fpm16 = {"gauss": 0, "trapezoid": 1, "zolotarev": 2}[args.contour]

    def step(precision=64, aspect=args.aspect, f16=fpm16, maxit=args.maxit):
        fpm = fk.feastinit()
        fpm[2], fpm[4], fpm[16], fpm[18] = NE, 40, f16, aspect
        return fk.feast_hip_hermitian(eng, A, B, EMIN, EMAX, M0, fpm, solver=args.solver, warm_start=True,
                                      inner_rtol=args.inner_rtol, solver_maxiter=maxit, preloaded=True,
                                      node_assignment="balanced", inner_precision=precision, column_groups="auto",
                                      Q0=Q0_dev, real_projection=True, reduced_solver=args.reduced_solver)

    def fence():
        eng.barrier()                         # library all-reduce over the communicator (no-op at N = 1)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.profile_reset()
    eng.profile_enable(os.environ.get("FEAST_BENCH_NOPROF") is None)   # sampled HIP-event timing (1 launch in 13)
    fence()
    t0 = time.perf_counter()
    results = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    elapsed = eng.max_over_ranks(elapsed)

    res = results[-1]
    ok = all(r.info == 0 and r.M == len(inside) for r in results)
    eig_err = float(np.abs(np.sort(res.lambda_) - inside).max()) if res.M == len(inside) else float("nan")
    # residual recomputed in fp64 on the host
    host_res = np.linalg.norm(A @ res.q - (B @ res.q) * res.lambda_, axis=0) / np.maximum(np.abs(res.lambda_), 1.0)
    max_res = float(host_res.max()) if res.M else float("nan")
    value = (sum(r.M for r in results) / elapsed) if ok else 0.0

    # ---- roofline of the dominant kernel ---------------------------------------------------------
    # candidates: the SpMM Y = (zB - A) X, the residual update (r -= a q, fused dots) and the direction/accumulator
    # kernel; the one with the largest share of the timed region is reported, the others kept under
    # "roofline_other".  Algorithmic bytes come from device-side counters of active (node, column) work per launch;
    # the average launch time from HIP events on the launch stream (1 launch in 13).
    N, nnz = A.shape[0], A.nnz
    src_hash = kernel_source_hash()
    pmc, pmc_file = {}, None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("pmc_traffic.json"):
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", name)))
            except Exception:
                continue
            if rec.get("kernel_source_hash") == src_hash:       # HBM counters of THIS tree's kernels only
                pmc, pmc_file = rec["kernels"], name
                break

    def roof(cls, kernel, alg_bytes):
        total_ms, launches = eng.profile_get(cls)
        if not (launches > 0 and total_ms > 0):
            return None
        avg_ms = total_ms / launches
        achieved = (alg_bytes / launches) / (avg_ms * 1e-3) / 1e9
        # every instantiation of the kernel the class launches (sum / plain, lazy-start first launch), weighted by launches
        tb = tl = 0
        for name, rec in pmc.items():
            if name.replace(" ", "").startswith("void" + kernel.split("<")[0] + "<") and ("cplx," in name or "<cplx" in name) and "cplxf" not in name:
                tb += rec["mean_hbm_bytes_per_launch"] * rec["launches"]
                tl += rec["launches"]
        traffic = int(tb / tl) if tl else None
        return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "peak_measured": HBM_MEASURED_GBS,
                "frac_of_measured_copy_peak": round(achieved / HBM_MEASURED_GBS["copy"], 4), "traffic": traffic,
                "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on kernel sources %s)"
                                   % (pmc_file, src_hash)) if traffic else None,
                "launches": int(launches), "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(alg_bytes / launches),
                "share_of_step": round(total_ms / (1e3 * elapsed), 3)}

    _, node_launches = eng.profile_get("spmm.node_launches")
    _, col_passes = eng.profile_get("spmm.column_passes")
    _, upd_cols = eng.profile_get("update.active_columns")
    matrix_bytes = nnz * (4 + 8 + 8) + 4 * (N + 1)            # col idx + A,B values (f64) + row pointers
    # (full-width panels over a real matrix go through the row-per-wave kernel; FH_SPMM_ROW=0 selects k_spmm)
    spmm_kernel = "k_spmm<cplx,double,64,false>" if os.environ.get("FH_SPMM_ROW") == "0" else "k_spmm_row<cplx,false,true>"
    spmm_roof = roof("spmm", spmm_kernel, node_launches * matrix_bytes + col_passes * N * 16)
    if spmm_roof:
        # the kernel streams all 64 columns of an active node (converged columns included): what it moves by design,
        # beside the algorithmic figure that counts active columns only
        spmm_roof["streamed_bytes_per_launch"] = int(node_launches * (matrix_bytes + 2 * 64 * N * 16) / max(spmm_roof["launches"], 1))
        spmm_roof["note"] = ("alg_bytes count ACTIVE columns; the kernel streams every column of an active node "
                             "(streamed_bytes); traffic / streamed is the cache excess (profiles/r04_dynamic_rows_experiment.txt)")
    cands = [spmm_roof]
    if args.solver == "cocg":
        _, v_launches = eng.profile_get("cocg_vec")
        if v_launches:
            # fused vector kernel, counted on the device per launch: a column that goes on iterating reads p, q, r and writes
            # r, p (5 passes; 4 in the first launch of a lazy start, which reads q and the shared source); a column on its last step only has its p read for the accumulator (1 pass); the shared
            # accumulator is read and written for the columns that stepped at any node
            _, cont_cols = eng.profile_get("update.continuing_columns")
            _, acc_cols = eng.profile_get("update.accumulator_columns")
            _, first_cols = eng.profile_get("update.first_launch_columns")      # lazy start's first launch: 4 passes, not 5
            cands.append(roof("cocg_vec", "k_fused_vec<cplx,64,true>",
                              (cont_cols * 5 - first_cols + max(upd_cols - cont_cols, 0) * 1 + acc_cols * 2) * N * 16))
        else:                                                     # FH_COCG_FUSED=0: the five-launch iteration
            _, p_launches = eng.profile_get("cocg_p")
            cands.append(roof("cocg_xr", "k_cocg_update<cplx,64>", upd_cols * 3 * N * 16))
            # direction update P = R + beta P (3 passes per active column) + the shared accumulator (read + write per launch)
            cands.append(roof("cocg_p", "k_cocg_p_sum<cplx,64>", upd_cols * 3 * N * 16 + p_launches * 2 * N * 64 * 16))
    else:
        cands.append(roof("bicg_xr", "k_xr_update<cplx,64>", upd_cols * 7 * N * 16))
    cands = sorted([r for r in cands if r], key=lambda r: -r["share_of_step"])
    classes = {}
    for cls in ("spmm", "cocg_vec", "cocg_xr", "cocg_p", "bicg_xr", "bicg_p", "bicg_s", "dot_finalize", "ortho", "gram", "accumulate", "ritz",
                "reduced_eig", "allreduce"):
        ms, n = eng.profile_get(cls)
        if n:
            classes[cls] = {"launches": int(n), "est_total_ms": round(ms, 2)}

    # per-node Krylov iterations of the last step, summed over its refinement loops (this rank's nodes)
    node_its = {}
    for nodes_l, per_loop in zip(res.stats.get("node_lists", []), res.stats.get("node_iterations", [])):
        for g, v in zip(nodes_l, per_loop):
            node_its[int(g)] = node_its.get(int(g), 0) + int(v)
    nr, rk, transport = eng.comm_size, eng.comm_rank, {0: "none", 1: "rccl", 2: "shm (ranks share a device)"}
    tr = eng.comm_transport() if hasattr(eng, "comm_transport") else 0
    contour_txt = "%s quadrature, ellipse ratio fpm[18]=%d (a/b = %.2f)" % (args.contour, args.aspect, args.aspect / 100.0)

    out = {
        "metric": "eigenpairs/sec + max residual, 16-node contour", "value": round(value, 3), "unit": "eigenpairs/s",
        "n_gpus": nr, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 2),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "cfg3: N=50000 sparse symmetric generalized (3-D Laplacian 50x40x25, B=I+0.1A), "
                               "interval (0,0.1775), 16 nodes (%s), M0=64, tol 1e-12" % contour_txt,
                   "solver": "batched %s, fp64, warm-started from Ritz pairs, inner rtol %g, <=%d its/loop"
                             % ("COCG (BiCG for the complex-symmetric shifted systems)" if args.solver == "cocg" else "BiCGStab",
                                args.inner_rtol, args.maxit),
                   "parallelism": "%d ranks = (node groups) x (column groups of >=16 RHS columns), near/far-axis nodes paired, "
                                  "1 packed all-reduce of Q_proj per loop inside the C ABI (transport: %s)" % (nr, transport.get(tr, "?"))},
        "eigenpairs": int(res.M), "expected_eigenpairs": int(len(inside)), "max_residual": max_res,
        "max_eigenvalue_error": eig_err, "loops": int(res.loop), "converged": bool(ok),
        "krylov_iterations_per_step": int(res.stats.get("krylov_iterations", 0)),
        "node_iterations_last_step": {str(k): node_its[k] for k in sorted(node_its)},
        "phase_seconds_last_step": {k: round(v, 4) for k, v in res.stats.get("phase_seconds", {}).items()},
        "solve_seconds_last_step": round(float(res.stats.get("solve_seconds", 0.0)), 4),
        "roofline": cands[0] if cands else None, "roofline_other": cands[1:], "kernel_classes": classes,
    }
    if rk == 0 and nr == 1 and not args.headline_only:
        # secondary measurements, none of them the headline value
        fence()
        step(32)
        fence()
        t1 = time.perf_counter()
        mixed = step(32)
        fence()
        dtm = time.perf_counter() - t1
        out["mixed_precision"] = {"value": round(mixed.M / dtm, 3) if mixed.info == 0 else 0.0, "unit": "eigenpairs/s",
                                  "note": "same solve with complex64 Krylov correction panels (not the headline value)",
                                  "max_residual_device": float(mixed.epsout), "loops": int(mixed.loop)}
        if not args.no_default_contour:
            step(64, 100, 0, 100)
            fence()
            t1 = time.perf_counter()
            dflt = step(64, 100, 0, 100)
            fence()
            dtd = time.perf_counter() - t1
            out["reference_default_contour"] = {
                "value": round(dflt.M / dtd, 3) if dflt.info == 0 and dflt.M == len(inside) else 0.0, "unit": "eigenpairs/s",
                "ms_per_step": round(1e3 * dtd, 2), "loops": int(dflt.loop), "krylov_iterations": int(dflt.stats.get("krylov_iterations", 0)),
                "note": "same solve on the reference's default contour (Gauss, circle fpm[18]=100, <=100 its/loop): round 1's configuration"}
            # the call a FeastKit.jl user makes: feast(A, B, (Emin, Emax); M0 = 64, fpm[2] = 16), nothing else set.  The whole
            # call is timed, matrix ingest and upload included; solver=:direct maps to the inexact Krylov configuration
            # and -- fpm[18] unset -- the driver picks the ellipse ratio itself (api.py, hip_backend contour_policy)
            import warnings
            fpm_d = fk.feastinit(); fpm_d[2] = NE
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                eng._problem_fp = None            # first call on a handle: ingest (union pattern, chunked rows) + upload inside
                fence()
                t1 = time.perf_counter()
                fk.feast(A, B, (EMIN, EMAX), M0=M0, fpm=fpm_d.copy(), engine=eng)
                fence()
                dt_first = time.perf_counter() - t1
                t1 = time.perf_counter()
                dc = fk.feast(A, B, (EMIN, EMAX), M0=M0, fpm=fpm_d.copy(), engine=eng)
                fence()
                dtc = time.perf_counter() - t1
            okc = dc.info == 0 and dc.M == len(inside)
            hres = np.linalg.norm(A @ dc.q - (B @ dc.q) * dc.lambda_, axis=0) / np.maximum(np.abs(dc.lambda_), 1.0) if okc else [float("nan")]
            out["reference_default_call"] = {
                "value": round(dc.M / dtc, 3) if okc else 0.0, "unit": "eigenpairs/s", "ms_per_call": round(1e3 * dtc, 2),
                "ms_first_call": round(1e3 * dt_first, 2), "value_first_call": round(dc.M / dt_first, 3) if okc else 0.0, "loops": int(dc.loop),
                "krylov_iterations": int(dc.stats.get("krylov_iterations", 0)), "max_residual": float(np.max(hres)),
                "fpm18_per_loop": dc.stats.get("contour_policy", {}).get("fpm18_per_loop"),
                "solver_substitution": dc.stats.get("solver_substitution"),
                "note": "fk.feast(A, B, (Emin, Emax), M0=64, fpm[2]=16) with every other keyword at its default.  ms_first_call: first call on the "
                        "handle, matrix ingest + upload inside the time; ms_per_call / value: a repeated call, the engine recognises the "
                        "resident matrices by a content fingerprint (a pass over their arrays, inside the time) and skips the ingest"}
            # the reference's own default for this input, a DIRECT solve per node (UMFPACK, src/sparse/feast_sparse.jl:339):
            # here the multifrontal LU on a nested-dissection tree (or, FH_MF=0, reverse Cuthill-McKee + blocked band LU), factors cached per node
            try:
                fpm_b = fk.feastinit(); fpm_b[2] = NE
                # a fresh ingest, so that the band plan (pattern scan + reverse Cuthill-McKee on the host) is made inside
                # the timed call like everything else a first direct call pays
                eng._problem_fp = None
                eng.set_problem(A, B)
                fence()
                t1 = time.perf_counter()
                eng.band_plan()
                dt_plan = time.perf_counter() - t1
                db = fk.feast(A, B, (EMIN, EMAX), M0=M0, fpm=fpm_b.copy(), engine=eng, solver="banded", keep_factors=True)
                fence()
                dtb = time.perf_counter() - t1
                t1 = time.perf_counter()
                db2 = fk.feast(A, B, (EMIN, EMAX), M0=M0, fpm=fpm_b.copy(), engine=eng, solver="banded", keep_factors=True)
                fence()
                dtb2 = time.perf_counter() - t1
                okb = db.info == 0 and db.M == len(inside) and db2.info == 0
                kl_b, ku_b, nbytes_b, _blk = eng.band_plan()
                plan_flops = eng.direct_plan_flops()
                bres = np.linalg.norm(A @ db.q - (B @ db.q) * db.lambda_, axis=0) / np.maximum(np.abs(db.lambda_), 1.0) if okb else [float("nan")]
                def split(r, total_s, plan_s=0.0):
                    ph = r.stats.get("phase_seconds", {})
                    rr = sum(ph.get(k, 0.0) for k in ("ortho", "project", "eig", "ritz"))
                    return {"plan": round(1e3 * plan_s, 1), "sweeps_wall": round(1e3 * ph.get("apply", 0.0), 1),
                            "sweeps_gpu": round(1e3 * float(r.stats.get("solve_seconds", 0.0)), 1), "rayleigh_ritz": round(1e3 * rr, 1),
                            "other": round(1e3 * (total_s - plan_s - ph.get("apply", 0.0) - rr), 1)}
                out["sparse_direct"] = {
                    "value": round(db.M / dtb, 3) if okb else 0.0, "unit": "eigenpairs/s", "ms_per_call": round(1e3 * dtb, 2),
                    "ms_cached_factors": round(1e3 * dtb2, 2), "loops": int(db.loop), "factorizations": int(db.stats.get("factorizations", 0)),
                    "split_ms_first_call": split(db, dtb, dt_plan), "split_ms_cached": split(db2, dtb2),
                    "split_note": "plan = pattern scan + reverse Cuthill-McKee + nested dissection / fronts / maps (host); sweeps_wall - sweeps_gpu "
                                  "= allocation of the factor slots (16 x GB_per_node, hipMalloc) and host staging; other = start subspace, "
                                  "eigenvector download, host checks",
                    "plan": {2: "multifrontal LU on a nested-dissection tree (batched dense fronts)", 1: "blocked band LU after reverse Cuthill-McKee",
                             0: "narrow band LU"}.get(int(_blk), "?"),
                    "flop_per_node": plan_flops, "band_flop_per_node": 8.0 * A.shape[0] * kl_b * (kl_b + ku_b),
                    "band": [int(kl_b), int(ku_b)], "GB_per_node": round(nbytes_b / 1e9, 3), "max_residual": float(np.max(bres)),
                    "note": "same solve with solver='banded', the library's sparse DIRECT solver: one LU of z B - A per node (the reference's "
                            "default for sparse input is a sparse LU per node, UMFPACK); ms_per_call factors all nodes, ms_cached_factors "
                            "repeats the call on the cached factors; `band` is the band the band LU would eliminate (FH_MF=0)"}
                eng.set_solver("cocg")
                eng.free_factors()
            except Exception as exc:
                out["sparse_direct_error"] = repr(exc)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, B, inside, res.lambda_)
        if not args.no_dense:
            try:
                out.update(dense_configs(fk, eng))
            except Exception as exc:          # the headline line must survive a failure of the extra measurements
                out["dense_error"] = repr(exc)
    if world > 1:
        import torch.distributed as dist
        eng.barrier()
        eng.comm_destroy()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
