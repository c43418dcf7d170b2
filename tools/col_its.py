import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import feastkit_jl_amd as fk
A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25, 0.1)
eng = fk.HipEngine(0)
eng.set_problem(A, B)
fpm = fk.feastinit(); fpm[2], fpm[4], fpm[16], fpm[18] = 16, 40, 0, 4000
trace = []
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.1775, 64, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2, solver_maxiter=50,
                           preloaded=True, real_projection=True, trace=trace)
print("info", r.info, "M", r.M, "loops", r.loop)
tot_used = tot_full = tot_t16 = tot_sorted16 = 0
for t in trace:
    ci = t["column_iterations"]            # nodes x active
    if ci is None: continue
    for e in range(ci.shape[0]):
        its = ci[e]
        mx = int(its.max())
        used = int(its.sum())
        # launches k = 1..mx: active columns at iteration k = count(its >= k)
        act = np.array([(its >= k).sum() for k in range(1, mx + 1)])
        # tile-masked in place (16-col tiles in original order): tiles with any active column
        t16 = 0
        for k in range(1, mx + 1):
            a = (its >= k)
            pad = np.zeros(64, bool); pad[:len(a)] = a
            t16 += 16 * int(pad.reshape(4, 16).any(axis=1).sum())
        s16 = int((np.ceil(act / 16) * 16).sum())
        tot_used += used; tot_full += 64 * mx; tot_t16 += t16; tot_sorted16 += s16
    print("loop", t["loop"], "node max its", [int(ci[e].max()) for e in range(ci.shape[0])], "active-col share %.2f" % (ci.sum() / (64.0 * ci.max(axis=1).sum())))
print("column-iterations used %d, full-width %d (%.3f), 16-col tiles in place %d (%.3f), compacted to 16-col tiles %d (%.3f)" % (
    tot_used, tot_full, tot_used / tot_full, tot_t16, tot_t16 / tot_full, tot_sorted16, tot_sorted16 / tot_full))
# per-column pattern of one slow node in a middle loop
t = trace[min(4, len(trace) - 1)]
print("loop", t["loop"], "node 15 column its:", t["column_iterations"][-1].tolist())
print("lambda order:", np.round(t["lambda"][:64], 4).tolist())
