"""Host-side product code (no GPU needed) through the C ABI, checked against the oracle:
merge scatter, normal-equation solve, LM / trust-region classes, spline evaluation / update."""
import ctypes as C
import os

import numpy as np
import pytest

from mba_vo_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))


@pytest.mark.parametrize("k", [2, 4])
def test_merge_host_matches_oracle(orc, mbavo, k):
    L = mbavo.load()
    rng = np.random.default_rng(1)
    F, N = 3, 6
    E = synth.packed_len(k)
    fb = rng.uniform(-1, 1, F * E)
    start = np.array([0, 1, 2 if k == 4 else 4], np.int32)
    n = 6 * N
    c1, H1, g1 = np.zeros(1), np.zeros(n * n), np.zeros(n)
    c2, H2, g2 = np.zeros(1), np.full(n * n, 3.0), np.full(n, 3.0)  # callee must zero H, g (merge...cpp:33-37)
    orc.lib().orc_merge_hessian_gradient_cost(F, k, orc.dp(fb), orc.ip(start), N, orc.dp(c1), orc.dp(H1), orc.dp(g1))
    assert L.mbavo_merge_host(F, k, mbavo.capi.dp(fb), mbavo.capi.ip(start), N, mbavo.capi.dp(c2), mbavo.capi.dp(H2), mbavo.capi.dp(g2)) == 0
    assert c1[0] == c2[0] and np.array_equal(H1, H2) and np.array_equal(g1, g2)
    c3 = np.zeros(1)
    assert L.mbavo_merge_host(F, k, mbavo.capi.dp(fb), mbavo.capi.ip(start), N, mbavo.capi.dp(c3), None, None) == 0
    assert c3[0] == c1[0]


def test_solve_normal_equation(orc, mbavo):
    L = mbavo.load()
    rng = np.random.default_rng(2)
    n = 48
    A = rng.uniform(-1, 1, (n, n))
    A = A.T @ A
    x = rng.uniform(-1, 1, n)
    b = A @ (-x)
    Af = A.ravel(order="F").copy()
    for solver in (0, 1):
        out, ref = np.zeros(n), np.zeros(n)
        assert L.mbavo_solve_normal_equation(mbavo.capi.dp(Af), mbavo.capi.dp(b), n, solver, mbavo.capi.dp(out)) == 0
        orc.lib().orc_solve_normal_equation(orc.dp(Af), orc.dp(b), n, solver, orc.dp(ref))
        assert np.abs(out - x).max() < 1e-8          # the harness' own bound (:1199-1201)
        assert np.abs(out - ref).max() < 1e-9
    assert L.mbavo_solve_normal_equation(mbavo.capi.dp(Af), mbavo.capi.dp(b), n, 7, mbavo.capi.dp(out)) != 0
    # LM-damped, badly scaled normal equations as the tracker produces them (t-block ~1e1, w-block ~1e4)
    J = rng.normal(size=(400, 24)) * np.r_[np.ones(12), 60 * np.ones(12)]
    Hd = J.T @ J
    Hd[np.diag_indices(24)] *= 1 + 1e-4
    g = J.T @ rng.normal(size=400)
    for solver in (0, 1):
        out, ref = np.zeros(24), np.zeros(24)
        L.mbavo_solve_normal_equation(mbavo.capi.dp(Hd.ravel(order="F").copy()), mbavo.capi.dp(g), 24, solver, mbavo.capi.dp(out))
        orc.lib().orc_solve_normal_equation(orc.dp(Hd.ravel(order="F").copy()), orc.dp(g), 24, solver, orc.dp(ref))
        assert np.abs(Hd @ out + g).max() < 1e-8 * np.abs(g).max()
        assert np.abs(out - ref).max() <= 1e-10 * np.abs(ref).max()
    # rank-deficient (knots no frame touches): SVD path returns the minimum-norm solution
    Hs = np.zeros((36, 36))
    Hs[:24, :24] = Hd
    gs = np.r_[g, np.zeros(12)]
    out, ref = np.zeros(36), np.zeros(36)
    L.mbavo_solve_normal_equation(mbavo.capi.dp(Hs.ravel(order="F").copy()), mbavo.capi.dp(gs), 36, 0, mbavo.capi.dp(out))
    orc.lib().orc_solve_normal_equation(orc.dp(Hs.ravel(order="F").copy()), orc.dp(gs), 36, 0, orc.dp(ref))
    assert np.abs(out[24:]).max() == 0 and np.abs(out - ref).max() <= 1e-10 * np.abs(ref).max()


def test_host_svd_solve_rank_threshold_is_eigens(mbavo):
    """The product's solver type 0 on systems with a prescribed spectrum (tests/test_oracle_harness_checks.py does the same for
    the oracle): singular values below epsilon * n * sigma_max are cut (Eigen's default JacobiSVD threshold,
    solve_normal_equation.h:20-26), everything above takes part -- against LAPACK's truncated pseudo-inverse at the same
    relative cut-off.  Well-conditioned systems take the LDL^T stand-in (ratio <= 1e8), the others the Jacobi SVD."""
    L = mbavo.load()
    rng = np.random.default_rng(21)
    n = 24
    eps = np.finfo(np.float64).eps
    for trial, (lows, rank) in enumerate([((), n), ((1e-19, 1e-18), n - 2), ((1e-12,), n), ((1e-12, 1e-20, 0.0), n - 2)]):
        Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        s = np.concatenate([np.logspace(0, -6, n - len(lows)), np.array(lows, float)]) * 37.0
        A = (Q * s) @ Q.T
        A = 0.5 * (A + A.T)
        b = rng.uniform(-1, 1, n)
        out = np.zeros(n)
        assert L.mbavo_solve_normal_equation(mbavo.capi.dp(A.ravel(order="F").copy()), mbavo.capi.dp(b), n, 0, mbavo.capi.dp(out)) == 0
        ref = -np.linalg.pinv(A, rcond=eps * n, hermitian=True) @ b
        kept = np.sort(s)[::-1][:rank]
        tol = 1e-13 * kept[0] / kept[-1] * max(1.0, np.abs(ref).max())
        assert np.abs(out - ref).max() < tol, (trial, np.abs(out - ref).max(), tol)


def test_lm_and_trust_region_follow_reference_script(mbavo):
    """Same script as tests/golden (outputs produced by the reference's own classes): bit-exact."""
    L = mbavo.load()
    lm, tr = L.mbavo_lm_new(), L.mbavo_tr_new(5)
    L.mbavo_tr_reset(tr, 100.0)
    for i in range(40):
        assert L.mbavo_tr_step_quality(tr, float(G["lm_c"][i]), float(G["lm_m"][i])) == G["tr_quality"][i]
        if G["lm_q"][i] > 0.5:
            L.mbavo_lm_step_accepted(lm, float(G["lm_q"][i]))
            L.mbavo_tr_step_accepted(tr, float(G["lm_c"][i]), float(G["lm_m"][i]))
        else:
            L.mbavo_lm_step_rejected(lm)
        if i == 25:
            L.mbavo_lm_reset(lm)
        assert L.mbavo_lm_get_radius(lm) == G["lm_radii"][i]
    L.mbavo_lm_delete(lm)
    L.mbavo_tr_delete(tr)


@pytest.mark.parametrize("k", [2, 4])
def test_spline_get_pose_and_plus(orc, mbavo, k):
    L = mbavo.load()
    kt, kR = synth.harness_spline()
    kt, kR = np.ascontiguousarray(kt.ravel()), np.ascontiguousarray(kR.ravel())
    nm = "c2" if k == 2 else "c4"
    for t in (0.26, 0.7, 1.23456, (7 - k) * 0.5 + 0.49):
        p, q, jt, jr = np.zeros(3), np.zeros(4), np.zeros(9 * k), np.zeros(12 * k)
        assert L.mbavo_spline_get_pose(k, 0.0, 0.5, mbavo.capi.dp(kt), mbavo.capi.dp(kR), 7, t, mbavo.capi.dp(p),
                                       mbavo.capi.dp(q), mbavo.capi.dp(jt), mbavo.capi.dp(jr)) == 0
        idx = L.mbavo_segment_start_index(t, 0.0, 0.5)
        u = t / 0.5 - idx
        po, qo, jto, jro = np.zeros(3), np.zeros(4), np.zeros(9 * k), np.zeros(12 * k)
        getattr(orc.lib(), "orc_%s_vec3" % nm)(orc.dp(kt[3 * idx:].copy()), u, orc.dp(po), orc.dp(jto))
        getattr(orc.lib(), "orc_%s_rot3" % nm)(orc.dp(kR[4 * idx:].copy()), u, orc.dp(qo), orc.dp(jro))
        assert np.abs(p - po).max() < 1e-13 and np.abs(q - qo).max() < 1e-15
        assert np.abs(jt - jto).max() < 1e-15 and np.abs(jr - jro).max() < 1e-13
    # out of the knot range: the reference asserts (Spline.h:232-234); here an error code
    assert L.mbavo_spline_get_pose(k, 0.0, 0.5, mbavo.capi.dp(kt), mbavo.capi.dp(kR), 7, 99.0, mbavo.capi.dp(p),
                                   mbavo.capi.dp(q), None, None) == -2
    # Plus_t / Plus_R (Spline.h:307-330)
    step = np.random.default_rng(3).normal(size=42) * 0.01
    ct, cR, ot, oR = np.zeros(21), np.zeros(28), np.zeros(21), np.zeros(28)
    assert L.mbavo_spline_plus(mbavo.capi.dp(kt), mbavo.capi.dp(kR), 7, mbavo.capi.dp(step), mbavo.capi.dp(ct), mbavo.capi.dp(cR)) == 0
    orc.lib().orc_plus_t(orc.dp(kt), orc.dp(step), 7, orc.dp(ot))
    orc.lib().orc_plus_R(orc.dp(kR), orc.dp(step[21:].copy()), 7, orc.dp(oR))
    assert np.array_equal(ct, ot) and np.abs(cR - oR).max() < 1e-15


def test_segment_start_index_golden(mbavo):
    L = mbavo.load()
    for t, i_ref in zip(G["seg_t"], G["seg_idx"]):
        assert L.mbavo_segment_start_index(float(t), 0.0, 0.5) == i_ref


def test_trust_region_degenerate_quotients_product(mbavo):
    """The product's TrustRegionStepEvaluator on the same degenerate rows (tests/test_oracle_golden.py: executed by the reference's
    compiled class): NaN where the reference's std::max returns NaN, the same bits elsewhere."""
    L = mbavo.load()
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))
    L.mbavo_tr_step_quality.restype = C.c_double
    for row, want in zip(G["tr_edge_script"], G["tr_edge_quality"]):
        tr = L.mbavo_tr_new(5)
        L.mbavo_tr_reset(tr, float(row[0]))
        if not np.isnan(row[1]):
            L.mbavo_tr_step_accepted(tr, float(row[1]), float(row[2]))
        got = L.mbavo_tr_step_quality(tr, float(row[3]), float(row[4]))
        L.mbavo_tr_delete(tr)
        assert (np.isnan(got) and np.isnan(want)) or got == want, (row, got, want)


def test_transformation_and_spline_frame_change_match_oracle(mbavo, orc):
    """Core::Transformation exp/log/*/inverse and SplineSE3::TransformTo of the product (host code, no device)
    against the oracle's restatement; tolerance 1e-14 (same closed forms, different quaternion-rotate grouping)."""
    import ctypes as C
    lib, L = mbavo.load(), orc.lib()
    dp = mbavo.capi.dp
    rng = np.random.default_rng(4)
    for scale in (0.0, 1e-12, 1e-4, 0.3, 1.0):
        a, b = rng.normal(0, 1, 6) * scale, rng.normal(0, 1, 6) * scale
        A, B, Ao, Bo = np.zeros(7), np.zeros(7), np.zeros(7), np.zeros(7)
        assert lib.mbavo_se3_exp(dp(a), dp(A)) == 0 and lib.mbavo_se3_exp(dp(b), dp(B)) == 0
        L.orc_se3_exp(orc.dp(a), orc.dp(Ao), orc.dp(Ao[3:])); L.orc_se3_exp(orc.dp(b), orc.dp(Bo), orc.dp(Bo[3:]))
        assert np.abs(A - Ao).max() < 1e-14 and np.abs(B - Bo).max() < 1e-14
        la, lo = np.zeros(6), np.zeros(6)
        assert lib.mbavo_se3_log(dp(A), dp(la)) == 0
        L.orc_se3_log(orc.dp(Ao), orc.dp(Ao[3:]), orc.dp(lo))
        assert np.abs(la - lo).max() < 1e-14 and np.abs(la - a).max() < 1e-12
        M, Mo, I, Io = np.zeros(7), np.zeros(7), np.zeros(7), np.zeros(7)
        lib.mbavo_transform_mul(dp(A), dp(B), dp(M)); L.orc_transform_mul(orc.dp(Ao), orc.dp(Bo), orc.dp(Mo))
        lib.mbavo_transform_inverse(dp(A), dp(I)); L.orc_transform_inverse(orc.dp(Ao), orc.dp(Io))
        assert np.abs(M - Mo).max() < 1e-14 and np.abs(I - Io).max() < 1e-14
    for k, N in ((2, 2), (4, 6)):
        kt, kR = synth.harness_spline(0.1, 0.2, N)
        kt, kR = np.ascontiguousarray(kt.ravel()), np.ascontiguousarray(kR.ravel())
        kt2, kR2 = kt.copy(), kR.copy()
        tgt = np.zeros(7)
        lib.mbavo_se3_exp(dp(np.array([0.3, -0.2, 0.1, 0.2, 0.1, -0.3])), dp(tgt))
        assert lib.mbavo_spline_transform_to(k, 0.0, 0.5, dp(kt), dp(kR), N, 0.2, dp(tgt[3:]), dp(tgt)) == 0
        L.orc_spline_transform_to(k, 0.0, 0.5, orc.dp(kt2), orc.dp(kR2), N, 0.2, orc.dp(tgt[3:]), orc.dp(tgt))
        assert np.abs(kt - kt2).max() < 1e-13 and np.abs(kR - kR2).max() < 1e-14
        assert lib.mbavo_spline_transform_to(k, 0.0, 0.5, dp(kt), dp(kR), N, 99.0, dp(tgt[3:]), dp(tgt)) == -2  # MBAVO_E_RANGE


def test_triangular_index_decode_closed_form():
    """pixel_math.h:tri_decode (packed upper-triangle index -> (row, col) through an fp32 square root and one correction
    step) restated in numpy float32 and compared with the row-by-row search of the reference's packing
    (compute_hessian_gradients_cost.cu:217-229) for every entry of every size the kernels use and some beyond."""
    def search(e, nd):
        i = 0
        while e >= nd - i:
            e -= nd - i
            i += 1
        return i, i + e

    def closed(e, nd):
        b = np.float32(2 * nd + 1)
        r = int((b - np.sqrt(np.float32(b * b - np.float32(8.0) * np.float32(e)), dtype=np.float32)) * np.float32(0.5))
        start = r * nd - r * (r - 1) // 2
        if start > e:
            r -= 1
            start = r * nd - r * (r - 1) // 2
        elif e - start >= nd - r:
            start += nd - r
            r += 1
        return r, r + (e - start)

    for nd in (1, 2, 12, 13, 24, 25, 36, 37, 49, 96, 97, 200):
        for e in range(nd * (nd + 1) // 2):
            assert closed(e, nd) == search(e, nd), (nd, e)


def test_packed_keyframe_holds_image_and_gradients_exactly(orc):
    """The packed keyframe word (mbavo_problem.grad_fp16 = 2: intensity | 2 dI/dx | 2 dI/dy, 8 + 9 + 9 bits) decodes to the u8
    image and to the oracle's gradient image (Gradient.h:16-75) bit for bit, extreme differences (+-255) and the zero border
    included; the halved blend on doubled differences rounds like the blend on the differences."""
    from mba_vo_amd import synth
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    img[5, 4:9] = [0, 7, 255, 9, 0]        # differences of +255 and -255 in x ...
    img[10:15, 20] = [0, 7, 255, 9, 0]     # ... and in y
    w = synth.pack_keyframe(img)
    assert w.dtype == np.uint32 and w.shape == img.shape
    I = (w & 0xff).astype(np.uint8)
    kx = ((w.astype(np.int64) << 47) >> 55).astype(np.int32)     # bits 8-16, sign-extended
    ky = (w.view(np.int32) >> 23)                                # bits 23-31, arithmetic shift
    g = np.zeros(img.shape + (2,), np.float32)
    mag = np.zeros(img.shape, np.float32)
    orc.lib().orc_image_gradients_u8(orc.u8p(img), img.shape[0], img.shape[1], orc.fp(g), orc.fp(mag))
    assert np.array_equal(I, img)
    assert np.array_equal(kx.astype(np.float32) * np.float32(0.5), g[..., 0]) and np.array_equal(ky.astype(np.float32) * np.float32(0.5), g[..., 1])
    assert kx.min() == -255 and kx.max() == 255 and ky.min() == -255 and ky.max() == 255
    assert (kx[0] == 0).all() and (kx[:, -1] == 0).all() and (ky[-1] == 0).all() and (ky[:, 0] == 0).all()
    # fp32 blend in the reference's order on the differences, and on the doubled differences then halved: the same bits
    wts = rng.random((1000, 4)).astype(np.float32)
    pick = rng.integers(0, kx.size, (1000, 4))
    k4 = kx.ravel()[pick].astype(np.float32)
    a = wts[:, 3] * (np.float32(0.5) * k4[:, 3])
    b = wts[:, 3] * k4[:, 3]
    for j in (2, 1, 0):
        a = a + wts[:, j] * (np.float32(0.5) * k4[:, j])
        b = b + wts[:, j] * k4[:, j]
    assert np.array_equal(a, np.float32(0.5) * b)
