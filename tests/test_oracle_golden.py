"""Pins the CPU oracle against vectors produced by EXECUTING the reference's own code
(tests/golden/make_golden.py -> ref_vectors.npz).  Bit-exact: both sides are compiled
without FMA contraction and perform the same operations."""
import ctypes as C
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))


def test_abi_sizes_match_reference():
    # Vector2d 24 B, Vector3d 32 B, VectorX<double,4> 40 B, VectorX<int,2> 12 B (Vector.h:11-18,72)
    assert list(G["abi_sizes"]) == [24, 32, 40, 12]


def test_quat_log_exp(orc):
    L = orc.lib()
    for i in range(64):
        t, J = np.zeros(3), np.zeros(12)
        L.orc_quat_log(orc.dp(np.ascontiguousarray(G["quat_in"][i])), orc.dp(t), orc.dp(J))
        assert np.array_equal(t, G["log_t"][i]) and np.array_equal(J, G["log_J"][i]), i
        q, Je = np.zeros(4), np.zeros(12)
        L.orc_quat_exp(orc.dp(np.ascontiguousarray(G["tangent_in"][i])), orc.dp(q), orc.dp(Je))
        assert np.array_equal(q, G["exp_q"][i]) and np.array_equal(Je, G["exp_J"][i]), i


@pytest.mark.parametrize("k,nm", [(2, "c2"), (4, "c4")])
def test_spline_functors(orc, k, nm):
    L = orc.lib()
    kt, kR, us = G["knots_t"], G["knots_R"], G["spline_us"]
    row = 0
    for idx in range(0, 7 - k + 1):
        for u in us:
            p, jt, q, jr = np.zeros(3), np.zeros(9 * k), np.zeros(4), np.zeros(12 * k)
            getattr(L, "orc_%s_vec3" % nm)(orc.dp(np.ascontiguousarray(kt[idx:].ravel())), float(u), orc.dp(p), orc.dp(jt))
            getattr(L, "orc_%s_rot3" % nm)(orc.dp(np.ascontiguousarray(kR[idx:].ravel())), float(u), orc.dp(q), orc.dp(jr))
            assert np.array_equal(p, G[nm + "_p"][row]) and np.array_equal(jt, G[nm + "_Jt"][row])
            assert np.array_equal(q, G[nm + "_q"][row]) and np.array_equal(jr, G[nm + "_JR"][row])
            row += 1


def test_identical_knots_series_branch(orc):
    q, jr = np.zeros(4), np.zeros(48)
    orc.lib().orc_c4_rot3(orc.dp(np.array([0, 0, 0, 1.0] * 4)), 0.37, orc.dp(q), orc.dp(jr))
    assert np.array_equal(q, G["ident_q"]) and np.array_equal(jr, G["ident_JR"])


def test_segment_index_truncates_toward_zero(orc):
    for t, i_ref, u_ref in zip(G["seg_t"], G["seg_idx"], G["seg_u"]):
        ii, u = C.c_int(), C.c_double()
        orc.lib().orc_spline_segment(float(t), 0.0, 0.5, C.byref(ii), C.cast(C.byref(u), orc.c_dp))
        assert ii.value == i_ref and u.value == u_ref
    assert list(G["seg_idx"][:3]) == [-1, 0, 0]  # (int)(-1.4) = -1, (int)(-0.4) = 0


@pytest.mark.parametrize("name", ["ramp", "noise"])
def test_pixel_intensity_and_bilinear(orc, mbavo, name):
    from mba_vo_amd import synth
    L = orc.lib()
    H, W = 480, 640
    img = synth.ramp_image(H, W) if name == "ramp" else synth.noise_image(H, W, seed=5)
    g = synth.image_gradients(img)
    q, t, D, xy = G["pi_%s_q" % name], G["pi_%s_t" % name], G["pi_%s_D" % name], G["pi_%s_xy" % name]
    n_ok = 0
    for i in range(len(D)):
        v, jac = np.zeros(1), np.zeros(7)
        ok = L.orc_pixel_intensity(orc.u8p(img), orc.fp(g), H, W, orc.dp(np.ascontiguousarray(q[i])),
                                   orc.dp(np.ascontiguousarray(t[i])), float(D[i]), 320.0, 320.0, 320.0, 240.0,
                                   float(xy[i, 0]), float(xy[i, 1]), orc.dp(v), orc.dp(jac))
        assert ok == G["pi_%s_ok" % name][i]
        if ok:
            n_ok += 1
            assert v[0] == G["pi_%s_val" % name][i]
            assert np.array_equal(jac, G["pi_%s_jac" % name][i])
    assert n_ok > 100
    bxy = G["bl_%s_xy" % name]
    for i in range(len(bxy)):
        o = np.zeros(3)
        ok = L.orc_bilinear(orc.u8p(img), orc.fp(g), H, W, float(bxy[i, 0]), float(bxy[i, 1]), orc.dp(o))
        assert ok == G["bl_%s_ok" % name][i]
        if ok:
            assert np.array_equal(o, G["bl_%s_val" % name][i])


def test_pyramid_and_gradients(orc, mbavo):
    from mba_vo_amd import synth
    L = orc.lib()
    src = np.ascontiguousarray(G["pyr_src"])
    H, W = src.shape
    cur = src
    for l, key in enumerate(["pyr_l1", "pyr_l2", "pyr_l3"], start=1):
        # the reference sizes level l as H0/2^l from level l-1 (ImagePyramid.h:72-73)
        Hl, Wl = H // 2 ** l, W // 2 ** l
        full = np.zeros((cur.shape[0] // 2, cur.shape[1] // 2), np.uint8)
        L.orc_pyramid_down_u8(orc.u8p(cur), cur.shape[0], cur.shape[1], orc.u8p(full))
        nxt = np.ascontiguousarray(full[:Hl, :Wl])
        assert np.array_equal(nxt, G[key])
        cur = nxt
    pyr = synth.pyramid(src, 4)
    for l, key in enumerate(["pyr_l1", "pyr_l2", "pyr_l3"], start=1):
        assert np.array_equal(pyr[l], G[key])
    g, mag = np.zeros((H, W, 2), np.float32), np.zeros((H, W), np.float32)
    L.orc_image_gradients_u8(orc.u8p(src), H, W, orc.fp(g), orc.fp(mag))
    assert np.array_equal(g, G["grad_xy"]) and np.array_equal(mag, G["grad_mag"])
    assert np.array_equal(synth.image_gradients(src), G["grad_xy"])


def test_lm_and_trust_region_script(orc):
    L = orc.lib()
    lm, tr = orc.OrcLm(), orc.OrcTr()
    L.orc_lm_init(C.byref(lm))
    L.orc_tr_init(C.byref(tr), 5)
    L.orc_tr_reset(C.byref(tr), 100.0)
    for i in range(40):
        ql = L.orc_tr_quality(C.byref(tr), float(G["lm_c"][i]), float(G["lm_m"][i]))
        assert ql == G["tr_quality"][i]
        if G["lm_q"][i] > 0.5:
            L.orc_lm_accepted(C.byref(lm), float(G["lm_q"][i]))
            L.orc_tr_accepted(C.byref(tr), float(G["lm_c"][i]), float(G["lm_m"][i]))
        else:
            L.orc_lm_rejected(C.byref(lm))
        if i == 25:
            L.orc_lm_reset(C.byref(lm))
        assert lm.radius == G["lm_radii"][i]


def test_trust_region_degenerate_quotients(orc):
    """StepQuality where its quotients are 0 / 0 or x / 0 (a tracker that has lost the scene evaluates cost 0 against cost 0 with a
    zero model change, tests/test_gpu_horizon.py k = 4): the reference's std::max hands back its FIRST argument when the comparison
    is unordered (trust_region_step_evaluator.cpp:74) -- rows executed by the reference's compiled class; NaN must be NaN, numbers
    the same bits.  (Round 6: the oracle returned the historical quotient where the relative one was NaN.)"""
    L = orc.lib()
    for row, want in zip(G["tr_edge_script"], G["tr_edge_quality"]):
        tr = orc.OrcTr()
        L.orc_tr_init(C.byref(tr), 5)
        L.orc_tr_reset(C.byref(tr), float(row[0]))
        if not np.isnan(row[1]):
            L.orc_tr_accepted(C.byref(tr), float(row[1]), float(row[2]))
        got = L.orc_tr_quality(C.byref(tr), float(row[3]), float(row[4]))
        assert (np.isnan(got) and np.isnan(want)) or got == want, (row, got, want)
    assert np.isnan(G["tr_edge_quality"][:2]).all() and G["tr_edge_quality"][2] == 2.0  # (the cases that tell std::max's argument order)


def test_oracle_matches_live_reference_when_present(orc):
    """Extra, denser comparison against oracle/_ref itself (only where it has been built)."""
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built on this machine")
    L = orc.lib()
    rng = np.random.default_rng(3)
    kR = np.ascontiguousarray(G["knots_R"].ravel())
    for _ in range(200):
        u = float(rng.uniform(0, 1))
        idx = int(rng.integers(0, 4))
        a, ja, b, jb = np.zeros(4), np.zeros(48), np.zeros(4), np.zeros(48)
        L.orc_c4_rot3(orc.dp(kR[idx * 4:]), u, orc.dp(a), orc.dp(ja))
        R.ref_c4_rot3(orc.dp(kR[idx * 4:]), u, orc.dp(b), orc.dp(jb))
        assert np.array_equal(a, b) and np.array_equal(ja, jb)


GS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_stage_vectors.npz"))


@pytest.mark.parametrize("name", ["k4", "k2", "k4_S3"])
def test_oracle_stages_reproduce_reference_on_whole_problems(orc, name):
    """Stage a3 (virtual poses + pose-to-knot Jacobians), stage a4 (patch centres) and stage a5 (per-pixel residual + 1 x 6k Jacobian) of the
    oracle against the REFERENCE's own per-sample code run over whole problems (tests/golden/make_stage_golden.py:
    spline functors, compute_pixel_intensity<double>, Core::MatrixMatrixMultiply compiled from /root/reference).
    Exact equality, including which pixels are invalid."""
    L = orc.lib()
    g = lambda k: GS["%s_%s" % (name, k)]
    S, F, K, P, k, N, H, W = (int(v) for v in g("in_scalars")[:8])
    t0, dt = float(g("in_scalars")[8]), float(g("in_scalars")[9])
    ref_img = np.ascontiguousarray(g("in_ref_img"))
    grad = np.zeros((H, W, 2), np.float32)
    L.orc_image_gradients_u8(orc.u8p(ref_img), H, W, orc.fp(grad), None)
    cur = [np.ascontiguousarray(c) for c in g("in_cur")]
    c64 = lambda a: np.ascontiguousarray(a, np.float64)
    kp_xy, kp_z, intr, cap, exp_t = c64(g("in_kp_xy")), c64(g("in_kp_z")), c64(g("in_intr")), c64(g("in_cap")), c64(g("in_exp_t"))
    kt, kR, pattern = c64(g("in_knots_t")), c64(g("in_knots_R")), np.ascontiguousarray(g("in_pattern"), np.int32)
    poses, Jt, JR = np.zeros(F * S * 7), np.zeros(F * S * 9 * k), np.zeros(F * S * 12 * k)
    L.orc_compute_virtual_camera_poses(S, F, orc.dp(cap), orc.dp(exp_t), k, t0, dt, orc.dp(kt), orc.dp(kR), orc.dp(poses),
                                       orc.dp(Jt), orc.dp(JR), None)
    assert np.array_equal(poses, g("out_poses")) and np.array_equal(Jt, g("out_J_t")) and np.array_equal(JR, g("out_J_R"))
    centres = np.zeros(F * K * 2)
    L.orc_compute_local_patches_xy(S, F, orc.dp(poses), orc.dp(kp_xy), orc.dp(kp_z), K, orc.dp(intr), orc.dp(centres))
    assert np.array_equal(centres, g("out_centres"))
    res, jac = np.zeros(F * K * P), np.zeros(F * K * P * 6 * k)
    cur_arr = (orc.c_u8p * F)(*[orc.u8p(c) for c in cur])
    L.orc_compute_pixel_jacobian_residual(orc.u8p(ref_img), orc.fp(grad), cur_arr, S, F, orc.dp(poses), k, orc.dp(Jt), orc.dp(JR),
                                          orc.dp(centres), orc.dp(kp_z), K, orc.ip(pattern), P, orc.dp(intr), H, W, orc.dp(res),
                                          orc.dp(jac))
    assert np.array_equal(res, g("out_residuals")) and np.array_equal(jac, g("out_jacobians"))
    assert np.count_nonzero(res) > 0.7 * res.size


def test_oracle_evaluation_matches_live_reference_stages_when_present(orc):
    """Larger random problems through oracle/_ref's stage drivers (only where it has been built): the whole
    evaluation of the oracle -- frame blocks included -- equals the one composed with the reference's per-sample code."""
    R = orc.ref()
    if R is None or not hasattr(R, "ref_compute_pixel_jacobian_residual"):
        pytest.skip("oracle/_ref (with stage drivers) not built on this machine")
    import scenes
    for kw in (dict(S=8, F=2, k=4, P=8, K=145), dict(S=4, F=1, k=2, P=8, K=145, huber=0.1),
               dict(H=96, W=128, S=16, F=1, k=4, P=1, kp="dense", margin=0)):
        sc = scenes.Scene(**kw)
        p, keep = sc.oracle_problem(orc)
        ro = orc.evaluate(p)
        rr = orc.stages_with_reference(dict(S=sc.S, F=sc.F, K=sc.K, P=sc.P, k=sc.k, N=sc.N, H=sc.H, W=sc.W, ref_img=sc.ref,
                                            ref_dIxy=sc.grad, cur_imgs=sc.cur, kp_xy=sc.kp_xy, kp_z=sc.kp_z, pattern=sc.pattern,
                                            intr=sc.intr, cap=sc.cap, exp_t=sc.exp, t0=sc.t0, dt=sc.dt, knots_t=sc.knots_t,
                                            knots_R=sc.knots_R, huber_a=sc.huber))
        assert np.array_equal(ro["frame_blocks"], rr["frame_blocks"])
