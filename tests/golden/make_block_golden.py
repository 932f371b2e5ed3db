"""Generates tests/golden/ref_block_vectors.npz: the three fixtures of SURVEY.md App. E that sit DOWNSTREAM of the
per-pixel rows -- `packed_blocks`, `merge_solve`, `lm_trace`.  The kernels that produce these in the reference
(compute_hessian_gradients_cost.cu:165-283, merge_hessian_gradient_cost.cpp, the tracker's loop) cannot be compiled
here (CUDA / Eigen / OpenCV), so every fixture is built from REFERENCE-EXECUTED inputs and cross-checked, inside this
script, against the independent formulas the reference's own harness uses for the same quantities:

  packed_blocks : per-pixel residuals and 1x6k Jacobians EXECUTED BY THE REFERENCE's per-sample code
                  (tests/golden/ref_stage_vectors.npz, made by oracle/_ref) -> per-patch packed blocks and per-frame
                  sums for Huber a in {0.1, 10, 1e32}, with and without an outlier mask: the oracle's stage 4 / 5
                  (restating ...cost.cu:179-238, :264-281), checked here against the harness' analytic Huber / outer
                  product (test/test_blur_aware_tracker_modules.cpp:958-982, tolerance 1e-8 / 1e-6 as there: its weight
                  is a / sqrt(x), the kernel's a / (sqrt(x) + 1e-8)) and plain sum (:1039-1050, 1e-8);
  merge_solve   : those frame blocks scattered with start indices {0, 1, 2} into N = 6 knots (harness :1105-1110) ->
                  H 36x36 column-major, g, cost: the oracle's merge, checked against the harness' block formula
                  (:1130-1153) accumulated pixel by pixel in numpy; plus the LDLT and minimum-norm (pseudo-inverse)
                  solutions from numpy.linalg with their residuals ||H x + g||;
  lm_trace      : the LM loop of blur_aware_direct_tracker.cpp:590-924 driven from Python with the REFERENCE's compiled
                  LevenbergMarquardtStrategy and TrustRegionStepEvaluator (oracle/_ref: ref_lm_*, ref_tr_*) deciding
                  radius / quality / acceptance, on one small synthetic pair per spline degree: per record (level,
                  iteration, kind, outliers, radius, evaluation cost, candidate cost, model change, quality).

Runs only in the build container (needs /root/reference for oracle/_ref); the file holds arrays only.

    python tests/golden/make_block_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import binding as B  # noqa: E402
import tracking  # noqa: E402

B.build()
L, R = B.lib(), B.ref()
assert R is not None and hasattr(R, "ref_lm_new"), "oracle/_ref is not built (needs /root/reference)"
stage = np.load(os.path.join(HERE, "ref_stage_vectors.npz"))
out = {}


def tri_index(nd):
    return [(i, j) for i in range(nd) for j in range(i, nd)]


def analytic_patch_block(res, jac, a, inv):
    """harness :958-982 for one patch: (cost, gradient, hessian) * inv, fp32 square roots as the harness writes them."""
    m = jac.shape[1]
    cost, g, H = 0.0, np.zeros(m), np.zeros((m, m))
    for r, J in zip(res, jac):
        x = 0.5 * r * r
        rho, w2 = x, 1.0
        if x > a * a:
            sx = float(np.sqrt(np.float32(x)))
            rho = 2 * a * sx - a * a
            w2 = a / sx
        cost += rho
        g += w2 * r * J
        H += w2 * np.outer(J, J)
    return cost * inv, g * inv, H * inv


# ---------------------------------------------------------------- packed_blocks + merge_solve
for name in ("k4", "k2"):
    S, F, K, P, k, N, H_, W_ = [int(v) for v in stage[name + "_in_scalars"][:8]]
    E, m = B.packed_len(k), 6 * k
    res = np.ascontiguousarray(stage[name + "_out_residuals"])          # reference-executed
    jac = np.ascontiguousarray(stage[name + "_out_jacobians"])
    inv = 1.0 / (K * F * P)
    rng = np.random.default_rng(99)
    mask = (rng.random(K) < 0.2).astype(np.uint8)
    out[name + "_mask"] = mask
    for a in (0.1, 10.0, 1e32):
        tag = "%s_a%g" % (name, a)
        pb = np.zeros(F * K * E)
        L.orc_compute_patch_cost_gradient_hessian(F, K, P, k, B.dp(res), B.dp(jac), a, inv, B.dp(pb))
        fb, fbm = np.zeros(F * E), np.zeros(F * E)
        L.orc_compute_frame_cost_gradient_hessian(F, K, k, B.dp(pb), 1, None, B.dp(fb))
        L.orc_compute_frame_cost_gradient_hessian(F, K, k, B.dp(pb), 1, B.u8p(mask), B.dp(fbm))
        pbr = pb.reshape(F * K, E)
        # cross-check every patch against the harness' analytic formula (its tolerances: 1e-8 cost, 1e-6 g / H)
        worst = [0.0, 0.0, 0.0]
        tri = tri_index(m + 1)
        for pi in range(F * K):
            c, g, Hm = analytic_patch_block(res[pi * P:(pi + 1) * P], jac.reshape(-1, m)[pi * P:(pi + 1) * P], a, inv)
            blk = pbr[pi]
            gk = blk[1:m + 1]
            Hk = np.zeros((m, m))
            for e, (i, j) in enumerate(tri):
                if i >= 1:
                    Hk[i - 1, j - 1] = Hk[j - 1, i - 1] = blk[e]
            # (relative to the block's magnitude: with a = 0.1 most pixels are in the Huber branch and the 1e-8 of the
            # kernel's denominator shows up as 1e-8 / sqrt(x) relative)
            worst = [max(worst[0], abs(c - blk[0])), max(worst[1], np.abs(g - gk).max() / max(1.0, np.abs(g).max())),
                     max(worst[2], np.abs(Hm - Hk).max() / max(1.0, np.abs(Hm).max()))]
        assert worst[0] <= 1e-8 and worst[1] <= 1e-6 and worst[2] <= 1e-6, (tag, worst)
        # frame sums against the plain sum (:1039-1050)
        plain = pbr.reshape(F, K, E).sum(1)
        assert np.abs(plain - fb.reshape(F, E)).max() <= 1e-8
        plain_m = (pbr.reshape(F, K, E) * (1 - mask)[None, :, None]).sum(1)
        assert np.abs(plain_m - fbm.reshape(F, E)).max() <= 1e-8
        out[tag + "_patch_blocks"], out[tag + "_frame_blocks"], out[tag + "_frame_blocks_masked"] = pbr, fb.reshape(F, E), fbm.reshape(F, E)
        print(tag, "patch blocks", pbr.shape, "max |oracle - harness formula|: cost %.1e g %.1e H %.1e" % tuple(worst))
    out[name + "_in_scalars"] = np.array([S, F, K, P, k, N, inv])

# merge_solve: 3 frames' blocks (two copies of the k4 frames + one more) with start indices 0, 1, 2 into N = 6 knots
k, m, E, N = 4, 24, 325, 6
S, F, K, P = [int(v) for v in stage["k4_in_scalars"][:4]]
res, jac = stage["k4_out_residuals"], stage["k4_out_jacobians"].reshape(-1, m)
fb2 = out["k4_a1e+32_frame_blocks"]                                     # Huber off, as in the harness' merge test
fb3 = np.ascontiguousarray(np.vstack([fb2[0], fb2[1], 0.5 * (fb2[0] + fb2[1])]))
start = np.array([0, 1, 2], np.int32)
n = 6 * N
cost, Hm, g = np.zeros(1), np.zeros(n * n), np.zeros(n)
L.orc_merge_hessian_gradient_cost(3, k, B.dp(fb3), B.ip(start), N, B.dp(cost), B.dp(Hm), B.dp(g))
# harness :1130-1153, pixel by pixel, for the three frames (frame 2 = the average of the other two)
inv = 1.0 / (K * F * P)
Hc, bc, cc = np.zeros((n, n)), np.zeros(n), 0.0
npx = K * P
for f, (wts, st) in enumerate([((1.0, 0.0), 0), ((0.0, 1.0), 1), ((0.5, 0.5), 2)]):
    for src, wt in enumerate(wts):
        if wt == 0.0:
            continue
        for i in range(src * npx, (src + 1) * npx):
            r, J = res[i], jac[i]
            cc += wt * 0.5 * r * r * inv
            b, Hl = wt * r * J * inv, wt * np.outer(J, J) * inv
            ia, ib = slice(3 * st, 3 * st + 12), slice(3 * (N + st), 3 * (N + st) + 12)
            bc[ia] += b[:12]; bc[ib] += b[12:]
            Hc[ia, ia] += Hl[:12, :12]; Hc[ib, ia] += Hl[12:, :12]; Hc[ia, ib] += Hl[:12, 12:]; Hc[ib, ib] += Hl[12:, 12:]
Hmat = Hm.reshape(n, n).T
assert abs(cc - cost[0]) <= 1e-10 * abs(cc) and np.abs(bc - g).max() <= 1e-10 * np.abs(bc).max() and np.abs(Hc - Hmat).max() <= 1e-10 * np.abs(Hc).max()
assert np.array_equal(Hmat, Hmat.T)
out["merge_frame_blocks"], out["merge_start"], out["merge_H_colmajor"], out["merge_g"], out["merge_cost"] = fb3, start, Hm, g, cost
# solutions.  (i) The LM-damped system H + 1e-4 diag(H) (what computeTrustRegionStep solves, :801-803): x = -Hd^-1 g
# (numpy LU; condition number stored).  (ii) Rank-deficient case A22: the same damped blocks merged into N = 7 knots,
# knot 6 untouched by any frame -> six exactly zero rows / columns; JacobiSVD::solve returns the minimum-norm solution
# (zero step on the untouched knot), LDLT does not apply.
Hd = Hmat + np.diag(np.diag(Hmat)) * 1e-4
x_d = -np.linalg.solve(Hd, g)
out["solve_H_damped_colmajor"], out["solve_x_damped"] = np.ascontiguousarray(Hd.T).ravel(), x_d
N7, n7 = 7, 42
idx = np.r_[np.arange(18), 21 + np.arange(18)]          # [t of knots 0..5 | w of knots 0..5] inside [t (21) | w (21)]
H7, g7 = np.zeros((n7, n7)), np.zeros(n7)
H7[np.ix_(idx, idx)] = Hd
g7[idx] = g
x7 = np.zeros(n7)
x7[idx] = x_d
x7p = -np.linalg.pinv(H7, rcond=1e-15, hermitian=True) @ g7
assert np.linalg.norm(x7p - x7) <= 1e-5 * np.linalg.norm(x7)       # rounding x cond(Hd) = 6e9
out["solve_H7_colmajor"], out["solve_g7"], out["solve_x7_minnorm"] = np.ascontiguousarray(H7.T).ravel(), g7, x7
out["solve_info"] = np.array([np.linalg.cond(Hd), np.linalg.norm(Hd @ x_d + g), np.linalg.norm(g)])
print("merge_solve: cost %.6g  cond(H + 1e-4 diag) %.2e  ||Hd x + g|| %.2e of ||g|| %.2e" % (
    cost[0], out["solve_info"][0], out["solve_info"][1], out["solve_info"][2]))


# ---------------------------------------------------------------- lm_trace
def lm_trace(sc, opts):
    """optimizePyramidLevel per level, the reference's LM strategy / step evaluator classes deciding (oracle/_ref)."""
    R.ref_lm_new.restype = C.c_void_p
    R.ref_tr_new.restype = C.c_void_p
    R.ref_lm_radius.restype = C.c_double
    R.ref_tr_quality.restype = C.c_double
    for fn, at in (("ref_lm_reset", [C.c_void_p]), ("ref_lm_rejected", [C.c_void_p]), ("ref_lm_accepted", [C.c_void_p, C.c_double]),
                   ("ref_lm_radius", [C.c_void_p]), ("ref_tr_reset", [C.c_void_p, C.c_double]),
                   ("ref_tr_quality", [C.c_void_p, C.c_double, C.c_double]), ("ref_tr_accepted", [C.c_void_p, C.c_double, C.c_double]),
                   ("ref_lm_delete", [C.c_void_p]), ("ref_tr_delete", [C.c_void_p])):
        getattr(R, fn).argtypes = at
    R.ref_tr_new.argtypes = [C.c_int]
    lm, tr = R.ref_lm_new(), R.ref_tr_new(opts["max_nonmono"])
    k, N, F = sc["k"], sc["N"], sc["F"]
    n, E = 6 * N, B.packed_len(k)
    kt, kR = sc["kt0"].ravel().copy(), sc["kR0"].ravel().copy()
    start = np.array([int((c - sc["t0"]) / sc["dt"]) for c in sc["cap"]], np.int32)
    trace = []
    eval_cost = 0.0
    for lv in range(len(sc["levels"]) - 1, -1, -1):
        Lv = sc["levels"][lv]
        K, P = Lv["kp_xy"].shape[0], Lv["pattern"].size // 2
        flags = np.zeros(max(K, 1), np.uint8)
        num_bad = [0]
        intr = sc["intr"] / (1 << lv)

        def evaluate(knt, knR, with_h):
            p, keep = B.make_problem(Lv["S"], F, K, P, k, N, Lv["H"], Lv["W"], Lv["ref"], Lv["grad"], Lv["cur"], Lv["kp_xy"], Lv["kp_z"],
                                     Lv["pattern"], intr, sc["cap"], sc["exp"], sc["t0"], sc["dt"], knt, knR, start, opts["huber_k"],
                                     outlier=flags, num_bad=num_bad[0])
            return B.evaluate(p, with_hessian=with_h)
        r = evaluate(kt, kR, True)
        eval_cost, Hm, g = r["cost"], np.ascontiguousarray(r["H"].T).ravel(), r["g"].copy()  # column-major
        R.ref_lm_reset(lm)
        R.ref_tr_reset(tr, eval_cost)
        trace.append((lv, 0, 0, 0, R.ref_lm_radius(lm), eval_cost, 0.0, 0.0, 0.0))
        it, abs_dec = 0, 1e10
        while True:
            it += 1
            if it > opts["max_num_iterations"] or abs_dec < opts["min_abs_cost_decrease"]:
                break
            irad = 1.0 / R.ref_lm_radius(lm)
            for i in range(n):
                Hm[i * n + i] += Hm[i * n + i] * irad
            step = np.zeros(n)
            L.orc_solve_normal_equation(B.dp(Hm), B.dp(g), n, opts["solver_type"], B.dp(step))
            Hmat = Hm.reshape(n, n).T
            model = -(float(g @ step) + 0.5 * float(step @ (Hmat @ step)))
            if model < 0:
                R.ref_lm_rejected(lm)
                trace.append((lv, it, 3, num_bad[0], R.ref_lm_radius(lm), eval_cost, 0.0, model, 0.0))
                continue
            ct, cR = np.zeros(3 * N), np.zeros(4 * N)
            L.orc_plus_t(B.dp(kt), B.dp(step), N, B.dp(ct))
            L.orc_plus_R(B.dp(kR), B.dp(np.ascontiguousarray(step[3 * N:])), N, B.dp(cR))
            rc = evaluate(ct, cR, False)
            cand = rc["cost"]
            abs_dec = eval_cost - cand
            q = R.ref_tr_quality(tr, cand, model)
            if q > opts["min_step_quality"] and cand < eval_cost:
                pc = rc["patch_blocks"][0, :, 0]
                keepm = pc >= 1e-8
                mu = pc[keepm].sum() / keepm.sum()
                var = ((pc[keepm] - mu) ** 2).sum() / keepm.sum()
                bad = np.abs(pc - mu) > opts["max_chi_square_error"] * float(np.sqrt(np.float32(var)))
                flags[:K][bad] = 1
                num_bad[0] = int(bad.sum())
                kt, kR = ct, cR
                r = evaluate(kt, kR, True)
                eval_cost, Hm, g = r["cost"], np.ascontiguousarray(r["H"].T).ravel(), r["g"].copy()
                R.ref_lm_accepted(lm, q)
                R.ref_tr_accepted(tr, eval_cost, model)
                trace.append((lv, it, 1, num_bad[0], R.ref_lm_radius(lm), eval_cost, cand, model, q))
                continue
            R.ref_lm_rejected(lm)
            trace.append((lv, it, 2, num_bad[0], R.ref_lm_radius(lm), eval_cost, cand, model, q))
    R.ref_lm_delete(lm)
    R.ref_tr_delete(tr)
    return np.array(trace), kt, kR


LM_CASES = {"lm_k4": dict(H=120, W=160, levels=3, S=8, k=4, seed=1), "lm_k2": dict(H=120, W=160, levels=3, S=8, k=2, seed=2)}
for name, kw in LM_CASES.items():
    sc = tracking.make_tracking_scene(B, **kw)
    tr, kt, kR = lm_trace(sc, tracking.OPTS)
    ro = tracking.run_oracle_tracker(B, sc, tracking.OPTS)      # the oracle's own C loop must agree with the driven one
    assert len(ro["trace"]) == len(tr), (len(ro["trace"]), len(tr))
    for a, b in zip(ro["trace"], tr):
        assert tuple(a[:4]) == tuple(int(v) for v in b[:4]), (a, b)
        assert abs(a[5] - b[5]) <= 1e-9 * abs(b[5]) and abs(a[4] - b[4]) <= 1e-6 * abs(b[4])
    out[name + "_kw"] = np.array(repr(kw))
    out[name + "_trace"], out[name + "_knots_t"], out[name + "_knots_R"] = tr, kt, kR
    print(name, "records", len(tr), "kinds", "".join(str(int(v)) for v in tr[:, 2]), "final cost %.6f" % tr[-1, 5])

np.savez_compressed(os.path.join(HERE, "ref_block_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "ref_block_vectors.npz"), os.path.getsize(os.path.join(HERE, "ref_block_vectors.npz")), "bytes")
