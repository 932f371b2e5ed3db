"""The analytic CPU cross-checks of the reference's module harness
(test/test_blur_aware_tracker_modules.cpp), restated with real pass/fail and applied
to the oracle.  The harness only prints; thresholds are the ones it prints against."""
import ctypes as C

import numpy as np
import pytest

from mba_vo_amd import synth


def _rotmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_compute_pixel_intensity_harness(orc):
    """:83-181 -- ramp image, ref pixel (20.5, 20.5): warped intensity == bilinear at the ref pixel (1e-4)
    and analytic 1x7 Jacobian ~ forward differences (eps 1e-6)."""
    L = orc.lib()
    H, W, fx, fy, cx, cy = 480, 640, 320.0, 320.0, 320.0, 240.0
    img = synth.ramp_image(H, W)
    g = synth.image_gradients(img)
    rng = np.random.default_rng(11)
    done = 0
    while done < 5:
        q = rng.normal(size=4) * 0.15
        q[3] = 1
        q /= np.linalg.norm(q)
        t = rng.uniform(-1, 1, 3)
        D = rng.uniform(5, 10)
        P3dr = np.array([(20.5 - cx) / fx * D, (20.5 - cy) / fy * D, D])
        Rm = _rotmat(q)
        P3dc = Rm.T @ P3dr - Rm.T @ t
        if P3dc[2] <= 0.1:
            continue
        cur = np.array([fx * P3dc[0] / P3dc[2] + cx, fy * P3dc[1] / P3dc[2] + cy])
        if not (0 <= cur[0] < W and 0 <= cur[1] < H):
            continue
        ref = np.zeros(3)
        assert L.orc_bilinear(orc.u8p(img), orc.fp(g), H, W, 20.5, 20.5, orc.dp(ref))
        val, Ja = np.zeros(1), np.zeros(7)
        assert L.orc_pixel_intensity(orc.u8p(img), orc.fp(g), H, W, orc.dp(q), orc.dp(t), D, fx, fy, cx, cy,
                                     cur[0], cur[1], orc.dp(val), orc.dp(Ja))
        assert abs(val[0] - ref[0]) < 1e-4
        eps, Jn = 1e-6, np.zeros(7)
        for i in range(3):
            t2 = t.copy(); t2[i] += eps
            v2 = np.zeros(1)
            L.orc_pixel_intensity(orc.u8p(img), orc.fp(g), H, W, orc.dp(q), orc.dp(t2), D, fx, fy, cx, cy, cur[0], cur[1], orc.dp(v2), None)
            Jn[i] = (v2[0] - val[0]) / eps
        for i in range(4):
            q2 = q.copy(); q2[i] += eps
            v2 = np.zeros(1)
            L.orc_pixel_intensity(orc.u8p(img), orc.fp(g), H, W, orc.dp(q2), orc.dp(t), D, fx, fy, cx, cy, cur[0], cur[1], orc.dp(v2), None)
            Jn[3 + i] = (v2[0] - val[0]) / eps
        # fp32 bilinear => FD noise ~1e-5/1e-6 * 255: compare loosely, as the harness does by eye
        assert np.allclose(Ja, Jn, rtol=0.05, atol=0.05 * np.abs(Ja).max() + 40.0)
        done += 1


def _harness_poses(orc, S=32, k=4):
    kt, kR = synth.harness_spline()
    cap = 0.25 + 0.5 * np.arange(4)
    exp = np.full(4, 0.1)
    F = 4
    poses, Jt, JR = np.zeros(F * S * 7), np.zeros(F * S * 9 * k), np.zeros(F * S * 12 * k)
    idx = np.zeros(F * S, np.int32)
    orc.lib().orc_compute_virtual_camera_poses(S, F, orc.dp(cap), orc.dp(exp), k, 0.0, 0.5, orc.dp(kt.ravel()),
                                               orc.dp(kR.ravel()), orc.dp(poses), orc.dp(Jt), orc.dp(JR), orc.ip(idx))
    return kt, kR, cap, exp, poses.reshape(F, S, 7), Jt.reshape(F, S, 3, 3 * k), JR.reshape(F, S, 4, 3 * k), idx.reshape(F, S)


def test_compute_virtual_camera_poses_harness(orc):
    """:183-342 -- sampled poses + Jacobians == SplineSE3::GetPose at t = cap - exp/2 + i*exp/(S-1), tol 1e-4."""
    S = 32
    kt, kR, cap, exp, poses, Jt, JR, idx = _harness_poses(orc, S)
    assert (idx == np.array([0, 1, 2, 3])[:, None]).all()
    for f, i in [(0, 0), (1, 7), (2, 31), (3, 16)]:
        t = cap[f] - exp[f] * 0.5 + i * exp[f] / (S - 1)
        ii, u = C.c_int(), C.c_double()
        orc.lib().orc_spline_segment(float(t), 0.0, 0.5, C.byref(ii), C.cast(C.byref(u), orc.c_dp))
        p, jt, q, jr = np.zeros(3), np.zeros(36), np.zeros(4), np.zeros(48)
        orc.lib().orc_c4_vec3(orc.dp(np.ascontiguousarray(kt[ii.value:].ravel())), u.value, orc.dp(p), orc.dp(jt))
        orc.lib().orc_c4_rot3(orc.dp(np.ascontiguousarray(kR[ii.value:].ravel())), u.value, orc.dp(q), orc.dp(jr))
        assert np.abs(poses[f, i, :3] - p).max() < 1e-4 and np.abs(poses[f, i, 3:] - q).max() < 1e-4
        assert np.abs(Jt[f, i].ravel() - jt).max() < 1e-4 and np.abs(JR[f, i].ravel() - jr).max() < 1e-4


def test_rotation_jacobian_is_derivative_of_right_perturbation(orc):
    """d q(u) / d w_j for R_j <- R_j * exp(w_j): analytic 4x12 vs central differences."""
    kt, kR = synth.harness_spline()
    L = orc.lib()
    u = 0.37
    q0, J = np.zeros(4), np.zeros(48)
    L.orc_c4_rot3(orc.dp(np.ascontiguousarray(kR[1:5].ravel())), u, orc.dp(q0), orc.dp(J))
    J = J.reshape(4, 12)
    eps = 1e-6
    for j in range(4):
        for a in range(3):
            d = np.zeros(12)
            qp, qm = np.zeros(4), np.zeros(4)
            for sgn, dst in ((1, qp), (-1, qm)):
                d[:] = 0
                d[3 * j + a] = sgn * eps
                cand = np.zeros(16)
                L.orc_plus_R(orc.dp(np.ascontiguousarray(kR[1:5].ravel())), orc.dp(d), 4, orc.dp(cand))
                L.orc_c4_rot3(orc.dp(cand), u, orc.dp(dst), None)
            assert np.abs((qp - qm) / (2 * eps) - J[:, 3 * j + a]).max() < 1e-7


def test_compute_local_patches_harness(orc):
    """:344-500 -- patch centre == re-projection at t_cap + exp/(S-1)/2 (pose index S/2)."""
    S = 32
    kt, kR, cap, exp, poses, _, _, _ = _harness_poses(orc, S)
    xy, z = synth.harness_keypoints()
    intr = np.array([320.0, 320.0, 320.0, 240.0])
    K = len(z)
    centres = np.zeros(4 * K * 2)
    orc.lib().orc_compute_local_patches_xy(S, 4, orc.dp(poses.ravel()), orc.dp(xy), orc.dp(z), K, orc.dp(intr), orc.dp(centres))
    centres = centres.reshape(4, K, 2)
    for f, i in [(0, 0), (2, 100), (3, 144)]:
        pose = poses[f, S // 2]
        Rm = _rotmat(pose[3:])
        P3dr = np.array([(xy[i, 0] - 320) / 320 * z[i], (xy[i, 1] - 240) / 320 * z[i], z[i]])
        Pc = Rm.T @ P3dr - Rm.T @ pose[:3]
        assert np.allclose(centres[f, i], [320 * Pc[0] / Pc[2] + 320, 320 * Pc[1] / Pc[2] + 240], atol=1e-8)
    t_mid = cap[2] - 0.05 + (S // 2) * 0.1 / (S - 1 + 1e-8)
    assert abs(t_mid - (cap[2] + 0.1 / (S - 1) * 0.5)) < 1e-9  # the harness' frame_t formula


def _pixel_stage(orc, kt, kR, S=32, k=4, F=4):
    L = orc.lib()
    H, W = 480, 640
    img = synth.ramp_image(H, W)
    g = synth.image_gradients(img)
    cap = np.ascontiguousarray(0.25 + 0.5 * np.arange(F))
    exp = np.full(F, 0.1)
    xy, z = synth.harness_keypoints()
    K, P = len(z), 8
    intr = np.array([320.0, 320.0, 320.0, 240.0])
    poses, Jt, JR = np.zeros(F * S * 7), np.zeros(F * S * 9 * k), np.zeros(F * S * 12 * k)
    L.orc_compute_virtual_camera_poses(S, F, orc.dp(cap), orc.dp(exp), k, 0.0, 0.5, orc.dp(np.ascontiguousarray(kt.ravel())),
                                       orc.dp(np.ascontiguousarray(kR.ravel())), orc.dp(poses), orc.dp(Jt), orc.dp(JR), None)
    centres = np.zeros(F * K * 2)
    L.orc_compute_local_patches_xy(S, F, orc.dp(poses), orc.dp(xy), orc.dp(z), K, orc.dp(intr), orc.dp(centres))
    res, jac = np.zeros(F * K * P), np.zeros(F * K * P * 6 * k)
    curs = (orc.c_u8p * F)(*[orc.u8p(img)] * F)
    L.orc_compute_pixel_jacobian_residual(orc.u8p(img), orc.fp(g), curs, S, F, orc.dp(poses), k, orc.dp(Jt), orc.dp(JR),
                                          orc.dp(centres), orc.dp(z), K, orc.ip(synth.PATTERN8), P, orc.dp(intr), H, W,
                                          orc.dp(res), orc.dp(jac))
    return dict(img=img, g=g, poses=poses.reshape(F, S, 7), centres=centres.reshape(F, K, 2), z=z,
                res=res.reshape(F, K, P), jac=jac.reshape(F, K, P, 6 * k), K=K, P=P)


def test_compute_pixel_jacobian_residual_harness(orc):
    """:502-895 -- residual of (frame 2, patch 100, pixel 1) == host loop over S calls of
    compute_pixel_intensity minus the un-interpolated current pixel; the 1x24 Jacobian
    matches finite differences on the knots (eps 1e-4, R <- R*exp(w)) through the full pipeline."""
    L = orc.lib()
    S = 32
    # small motion so that the ramp image's mod-255 wrap is not crossed by the FD perturbation
    kt, kR = synth.harness_spline(trans_scale=0.02, rot_scale=0.2)
    st = _pixel_stage(orc, kt, kR, S)
    f, kp, px = 2, 100, 1
    c = st["centres"][f, kp]
    x = float(int(c[0] + synth.PATTERN8[2 * px]))
    y = float(int(c[1] + synth.PATTERN8[2 * px + 1]))
    acc = 0.0
    for i in range(S):
        v = np.zeros(1)
        pose = np.ascontiguousarray(st["poses"][f, i])
        assert L.orc_pixel_intensity(orc.u8p(st["img"]), None, 480, 640, orc.dp(pose[3:].copy()), orc.dp(pose[:3].copy()),
                                     float(st["z"][kp]), 320.0, 320.0, 320.0, 240.0, x, y, orc.dp(v), None)
        acc += v[0] / np.float32(S)
    cpu_res = acc - float(st["img"][int(y), int(x)])
    assert abs(cpu_res - st["res"][f, kp, px]) < 1e-9
    # finite differences on the 4 knots of frame 2 (start index 2)
    eps, Ja = 1e-4, st["jac"][f, kp, px]
    Jn = np.zeros(24)
    for i in range(24):
        kt2, kR2 = kt.copy(), kR.copy()
        if i < 12:
            kt2[2 + i // 3, i % 3] += eps
        else:
            d = np.zeros(12)
            d[i - 12] = eps
            cand = np.zeros(16)
            L.orc_plus_R(orc.dp(np.ascontiguousarray(kR[2:6].ravel())), orc.dp(d), 4, orc.dp(cand))
            kR2[2:6] = cand.reshape(4, 4)
        st2 = _pixel_stage(orc, kt2, kR2, S)
        # compare at the SAME integer pixel: recompute only if the truncated location did not move
        Jn[i] = (st2["res"][f, kp, px] - st["res"][f, kp, px]) / eps
    ok = np.isclose(Ja, Jn, rtol=0.02, atol=0.02 * np.abs(Ja).max())
    assert ok.sum() >= 22, (Ja, Jn)  # fp32 interpolation noise / eps leaves 3-4 digits, as the survey measured


def test_compute_patch_cost_gradient_hessian_harness(orc):
    """:897-1011 -- random r, J (5 x 145 x 8 pixels, k=4), huber 0.1, inv 1: patch 96 vs explicit
    rho, rho'*r*J, rho'*J*J^T; tol 1e-8 cost, 1e-6 g/H."""
    rng = np.random.default_rng(5)
    F, K, P, k = 5, 145, 8, 4
    res = rng.uniform(-1, 1, F * K * P)
    jac = rng.uniform(-1, 1, F * K * P * 24)
    blocks = np.zeros(F * K * 325)
    a = 0.1
    orc.lib().orc_compute_patch_cost_gradient_hessian(F, K, P, k, orc.dp(res), orc.dp(jac), a, 1.0, orc.dp(blocks))
    for patch in (96, 0, F * K - 1):
        blk = blocks[patch * 325:(patch + 1) * 325]
        cost, gvec, Hm = 0.0, np.zeros(24), np.zeros((24, 24))
        for i in range(P):
            r = res[patch * P + i]
            J = jac[(patch * P + i) * 24:(patch * P + i + 1) * 24]
            x = 0.5 * r * r
            rho, d = x, 1.0
            if x > a * a:
                sx = float(np.sqrt(np.float32(x)))  # the harness uses sqrtf too (:971-972)
                rho = 2 * a * sx - a * a
                d = a / sx
            cost += rho
            gvec += d * r * J
            Hm += d * np.outer(J, J)
        assert abs(blk[0] - cost) < 1e-8
        assert np.abs(blk[1:25] - gvec).max() < 1e-6
        iu = np.triu_indices(24)
        assert np.abs(blk[25:] - Hm[iu]).max() < 1e-6


def test_compute_frame_cost_gradient_hessian_harness(orc):
    """:1013-1058 -- random 6 x 145 x 325: per-frame sums == column sums, tol 1e-8; outlier mask skips patches."""
    rng = np.random.default_rng(6)
    F, K, E = 6, 145, 325
    blocks = rng.uniform(-1, 1, F * K * E)
    fb = np.zeros(F * E)
    orc.lib().orc_compute_frame_cost_gradient_hessian(F, K, 4, orc.dp(blocks), 1, None, orc.dp(fb))
    assert np.abs(fb.reshape(F, E) - blocks.reshape(F, K, E).sum(1)).max() < 1e-8
    flags = (rng.random(K) < 0.2).astype(np.uint8)
    orc.lib().orc_compute_frame_cost_gradient_hessian(F, K, 4, orc.dp(blocks), 1, orc.u8p(flags), orc.dp(fb))
    assert np.abs(fb.reshape(F, E) - blocks.reshape(F, K, E)[:, flags == 0].sum(1)).max() < 1e-8
    fb2 = np.full(F * E, 7.0)
    orc.lib().orc_compute_frame_cost_gradient_hessian(F, K, 4, orc.dp(blocks), 0, None, orc.dp(fb2))
    fb2 = fb2.reshape(F, E)
    assert np.abs(fb2[:, 0] - blocks.reshape(F, K, E)[:, :, 0].sum(1)).max() < 1e-8 and (fb2[:, 1:] == 7.0).all()


def _merge_reference(res, jac, F, K, P, N, start):
    n = 6 * N
    Hc, bc, cost = np.zeros((n, n)), np.zeros(n), 0.0
    for i in range(F * K * P):
        r, J = res[i], jac[i * 24:(i + 1) * 24]
        s = start[i // (K * P)]
        cost += 0.5 * r * r
        b, Hh = r * J, np.outer(J, J)
        t0, r0 = s * 3, (N + s) * 3
        bc[t0:t0 + 12] += b[:12]
        bc[r0:r0 + 12] += b[12:]
        Hc[t0:t0 + 12, t0:t0 + 12] += Hh[:12, :12]
        Hc[r0:r0 + 12, t0:t0 + 12] += Hh[12:, :12]
        Hc[t0:t0 + 12, r0:r0 + 12] += Hh[:12, 12:]
        Hc[r0:r0 + 12, r0:r0 + 12] += Hh[12:, 12:]
    return cost, Hc, bc


def test_merge_hessian_gradient_cost_harness(orc):
    """:1060-1181 -- 3 frames, start idx 0,1,2, N = 6 knots, huber 1e32: merged H/g == explicit block scatter, tol 1e-4."""
    rng = np.random.default_rng(7)
    F, K, P, k, N = 3, 145, 8, 4, 6
    res = rng.uniform(-1, 1, F * K * P)
    jac = rng.uniform(-1, 1, F * K * P * 24)
    blocks, fb = np.zeros(F * K * 325), np.zeros(F * 325)
    L = orc.lib()
    L.orc_compute_patch_cost_gradient_hessian(F, K, P, k, orc.dp(res), orc.dp(jac), 1e32, 1.0, orc.dp(blocks))
    L.orc_compute_frame_cost_gradient_hessian(F, K, k, orc.dp(blocks), 1, None, orc.dp(fb))
    start = np.array([0, 1, 2], np.int32)
    cost, H, g = np.zeros(1), np.zeros(36 * 36), np.zeros(36)
    L.orc_merge_hessian_gradient_cost(F, k, orc.dp(fb), orc.ip(start), N, orc.dp(cost), orc.dp(H), orc.dp(g))
    c_ref, H_ref, g_ref = _merge_reference(res, jac, F, K, P, N, start)
    Hm = H.reshape(36, 36).T
    assert abs(cost[0] - c_ref) < 1e-4 and np.abs(g - g_ref).max() < 1e-4 and np.abs(Hm - H_ref).max() < 1e-4
    assert np.array_equal(Hm, Hm.T)


def test_solve_normal_equation_harness_and_properties(orc):
    """:1183-1208 -- random SPD 48x48, LDLT, known x (1e-8).  JacobiSVD solve is not pinned by the
    reference's tests: checked through A x = -b, and the minimum-norm property on a rank-deficient H."""
    rng = np.random.default_rng(8)
    n = 48
    A = rng.uniform(-1, 1, (n, n))
    A = A.T @ A
    x = rng.uniform(-1, 1, n)
    b = A @ (-x)
    for solver in (1, 0):
        out = np.zeros(n)
        orc.lib().orc_solve_normal_equation(orc.dp(np.asfortranarray(A).ravel(order="F").copy()), orc.dp(b), n, solver, orc.dp(out))
        assert np.abs(out - x).max() < 1e-8, solver
    # rank-deficient: two knots untouched by any frame -> zero rows/cols; SVD gives the pseudo-inverse solution
    m = 24
    B_ = rng.uniform(-1, 1, (m, m))
    Hs = np.zeros((36, 36))
    Hs[:m, :m] = B_.T @ B_
    g = np.zeros(36)
    g[:m] = rng.uniform(-1, 1, m)
    out = np.zeros(36)
    rank = orc.lib().orc_solve_normal_equation(orc.dp(Hs.ravel(order="F").copy()), orc.dp(g), 36, 0, orc.dp(out))
    assert rank == m
    assert np.abs(out - (-np.linalg.pinv(Hs) @ g)).max() < 1e-7
    assert np.abs(out[m:]).max() == 0.0


def test_so3_exp_is_rodrigues(orc):
    """Sophus::SO3d::exp is third-party and unpinned: closed-form check against the Rodrigues formula."""
    rng = np.random.default_rng(9)
    for sc in (1.0, 1e-3, 1e-9, 0.0):
        w = rng.normal(size=3) * sc
        q = np.zeros(4)
        orc.lib().orc_so3_exp(orc.dp(w), orc.dp(q))
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        Rr = np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
        assert np.abs(_rotmat(q) - Rr).max() < 1e-12
        assert abs(np.linalg.norm(q) - 1) < 1e-14


def test_svd_solve_rank_threshold_is_eigens(orc):
    """solve_normal_equation.h:10-35 solves with Eigen::JacobiSVD at its DEFAULT threshold: singular values up to
    epsilon * diagSize * sigma_max count as zero (Eigen/src/SVD/SVDBase.h: threshold(), rank()), everything above takes part
    with its full 1 / sigma.  Eigen is not in this image; what its solve() computes is the truncated pseudo-inverse, which
    LAPACK's SVD gives independently: symmetric systems with a prescribed spectrum -- well separated from the threshold on
    either side, so that the rank is unambiguous -- against numpy's pinv at the same relative cut-off, and the rank returned."""
    rng = np.random.default_rng(21)
    n = 24
    eps = np.finfo(np.float64).eps
    for trial, (lows, want_rank) in enumerate([((), n), ((1e-19, 1e-18), n - 2), ((1e-12,), n), ((1e-12, 1e-20, 0.0), n - 2)]):
        Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        s = np.concatenate([np.logspace(0, -6, n - len(lows)), np.array(lows, float)]) * 37.0
        A = (Q * s) @ Q.T
        A = 0.5 * (A + A.T)
        b = rng.uniform(-1, 1, n)
        out = np.zeros(n)
        rank = orc.lib().orc_solve_normal_equation(orc.dp(A.ravel(order="F").copy()), orc.dp(b), n, 0, orc.dp(out))
        assert rank == want_rank, (trial, rank)
        ref = -np.linalg.pinv(A, rcond=eps * n, hermitian=True) @ b
        # the kept directions' error scales with 1 / sigma_min(kept) relative to sigma_max
        kept = np.sort(s)[::-1][:want_rank]
        tol = 1e-13 * kept[0] / kept[-1] * max(1.0, np.abs(ref).max())
        assert np.abs(out - ref).max() < tol, (trial, np.abs(out - ref).max(), tol)
