"""-m gpu: the reference's five launchers, one by one, through the C ABI against the oracle on the
module-harness inputs (test/test_blur_aware_tracker_modules.cpp).  fp64 chains: 1e-9 relative
(observed ~1e-15); integer / byte outputs exact."""
import ctypes as C

import numpy as np
import pytest

from mba_vo_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _vec2d(xy):
    v = np.zeros((xy.shape[0], 3), np.float64)
    v.view(np.int32)[:, 0] = 2
    v[:, 1:] = xy
    return v


@pytest.mark.parametrize("k,S", [(4, 32), (2, 8), (4, 1)])
def test_stage_by_stage(orc, mbavo, gpu_ctx, k, S):
    import torch
    L, O = mbavo.load(), orc.lib()
    H, W, F, P = 480, 640, 4, 8
    kt, kR = synth.harness_spline(trans_scale=0.01, rot_scale=0.1)
    cap = np.ascontiguousarray(0.25 + 0.5 * np.arange(F))
    exp = np.full(F, 0.1)
    img = synth.noise_image(H, W, seed=3)
    grad = synth.image_gradients(img)
    cur = np.ascontiguousarray(np.roll(img, (1, -2), (0, 1)))
    xy, z = synth.harness_keypoints()
    z = z / 4.0
    K = len(z)
    intr = np.array([320.0, 320.0, 320.0, 240.0])
    hw = np.array([H, W], np.int32)
    nJt, nJR = 9 * k, 12 * k
    # ---- oracle
    o_poses, o_Jt, o_JR = np.zeros(F * S * 7), np.zeros(F * S * nJt), np.zeros(F * S * nJR)
    O.orc_compute_virtual_camera_poses(S, F, orc.dp(cap), orc.dp(exp), k, 0.0, 0.5, orc.dp(kt.ravel()), orc.dp(kR.ravel()),
                                       orc.dp(o_poses), orc.dp(o_Jt), orc.dp(o_JR), None)
    o_c = np.zeros(F * K * 2)
    O.orc_compute_local_patches_xy(S, F, orc.dp(o_poses), orc.dp(xy), orc.dp(z), K, orc.dp(intr), orc.dp(o_c))
    o_res, o_jac = np.zeros(F * K * P), np.zeros(F * K * P * 6 * k)
    curs = (orc.c_u8p * F)(*[orc.u8p(cur)] * F)
    O.orc_compute_pixel_jacobian_residual(orc.u8p(img), orc.fp(grad), curs, S, F, orc.dp(o_poses), k, orc.dp(o_Jt),
                                          orc.dp(o_JR), orc.dp(o_c), orc.dp(z), K, orc.ip(synth.PATTERN8), P,
                                          orc.dp(intr), H, W, orc.dp(o_res), orc.dp(o_jac))
    E = synth.packed_len(k)
    inv = 1.0 / (K * F * P)
    o_pb = np.zeros(F * K * E)
    O.orc_compute_patch_cost_gradient_hessian(F, K, P, k, orc.dp(o_res), orc.dp(o_jac), 10.0, inv, orc.dp(o_pb))
    flags = (np.random.default_rng(1).random(K) < 0.1).astype(np.uint8)
    o_fb = np.zeros(F * E)
    O.orc_compute_frame_cost_gradient_hessian(F, K, k, orc.dp(o_pb), 1, orc.u8p(flags), orc.dp(o_fb))
    # ---- HIP
    d_cap, d_exp, d_kt, d_kR = _t(cap), _t(exp), _t(kt.ravel()), _t(kR.ravel())
    d_poses = torch.zeros(F * S * 7, dtype=torch.float64, device="cuda:0")
    d_Jt = torch.zeros(F * S * nJt, dtype=torch.float64, device="cuda:0")
    d_JR = torch.zeros(F * S * nJR, dtype=torch.float64, device="cuda:0")
    assert L.mbavo_compute_virtual_camera_poses(S, F, d_cap.data_ptr(), d_exp.data_ptr(), k, 0.0, 0.5, d_kt.data_ptr(),
                                                d_kR.data_ptr(), d_poses.data_ptr(), d_Jt.data_ptr(), d_JR.data_ptr()) == 0
    assert _rel(d_poses.cpu().numpy(), o_poses) < RTOL
    assert np.abs(d_Jt.cpu().numpy() - o_Jt).max() < 1e-14
    assert _rel(d_JR.cpu().numpy(), o_JR) < RTOL
    d_kp, d_z = _t(_vec2d(xy)), _t(z)
    d_c = torch.zeros(F * K * 3, dtype=torch.float64, device="cuda:0")
    assert L.mbavo_compute_local_patches_xy(S, F, d_poses.data_ptr(), d_kp.data_ptr(), d_z.data_ptr(), K,
                                            mbavo.capi.dp(intr), mbavo.capi.ip(hw), d_c.data_ptr()) == 0
    c = d_c.cpu().numpy().reshape(F * K, 3)
    assert (c.view(np.int32)[:, 0] == 2).all()          # nDim header of Core::Vector2d
    assert _rel(c[:, 1:].ravel(), o_c) < RTOL
    d_img, d_grad, d_cur = _t(img), _t(grad), _t(cur)
    d_curs = torch.tensor([d_cur.data_ptr()] * F, dtype=torch.int64, device="cuda:0")
    d_pat = _t(synth.PATTERN8)
    d_res = torch.zeros(F * K * P, dtype=torch.float64, device="cuda:0")
    d_jac = torch.zeros(F * K * P * 6 * k, dtype=torch.float64, device="cuda:0")
    assert L.mbavo_compute_pixel_jacobian_residual(d_img.data_ptr(), d_grad.data_ptr(), d_curs.data_ptr(), S, F,
                                                   d_poses.data_ptr(), k, d_Jt.data_ptr(), d_JR.data_ptr(), d_c.data_ptr(),
                                                   d_z.data_ptr(), K, d_pat.data_ptr(), P, mbavo.capi.dp(intr),
                                                   mbavo.capi.ip(hw), d_res.data_ptr(), d_jac.data_ptr()) == 0
    # residuals carry the fp32 bilinear island: 2e-5 absolute on intensities, observed exact
    assert np.abs(d_res.cpu().numpy() - o_res).max() < 2e-5
    assert _rel(d_jac.cpu().numpy(), o_jac) < RTOL
    # cost-only mode of the same launcher: Jacobian output pointer null
    d_res2 = torch.zeros_like(d_res)
    assert L.mbavo_compute_pixel_jacobian_residual(d_img.data_ptr(), d_grad.data_ptr(), d_curs.data_ptr(), S, F,
                                                   d_poses.data_ptr(), k, None, None, d_c.data_ptr(), d_z.data_ptr(), K,
                                                   d_pat.data_ptr(), P, mbavo.capi.dp(intr), mbavo.capi.ip(hw),
                                                   d_res2.data_ptr(), None) == 0
    assert torch.equal(d_res, d_res2)
    d_pb = torch.zeros(F * K * E, dtype=torch.float64, device="cuda:0")
    # feed the ORACLE's rows so this stage is compared on identical inputs
    assert L.mbavo_compute_patch_cost_gradient_hessian(F, K, P, k, _t(o_res).data_ptr(), _t(o_jac).data_ptr(), 10.0, inv,
                                                       d_pb.data_ptr()) == 0
    assert _rel(d_pb.cpu().numpy(), o_pb) < RTOL
    d_fb = torch.zeros(F * E, dtype=torch.float64, device="cuda:0")
    assert L.mbavo_compute_frame_cost_gradient_hessian(F, K, k, _t(o_pb).data_ptr(), 1, _t(flags).data_ptr(), d_fb.data_ptr()) == 0
    assert np.array_equal(d_fb.cpu().numpy(), o_fb)     # same 256-lane + tree order: bit-exact
    # merge: D2H + scatter
    N = 7
    start = np.arange(F, dtype=np.int32)[:F] if k == 4 else np.arange(F, dtype=np.int32)
    n = 6 * N
    c1, H1, g1 = np.zeros(1), np.zeros(n * n), np.zeros(n)
    c2, H2, g2 = np.zeros(1), np.zeros(n * n), np.zeros(n)
    assert L.mbavo_merge_hessian_gradient_cost(F, k, d_fb.data_ptr(), mbavo.capi.ip(start), N, mbavo.capi.dp(c1),
                                               mbavo.capi.dp(H1), mbavo.capi.dp(g1)) == 0
    O.orc_merge_hessian_gradient_cost(F, k, orc.dp(o_fb), orc.ip(start), N, orc.dp(c2), orc.dp(H2), orc.dp(g2))
    assert c1[0] == c2[0] and np.array_equal(H1, H2) and np.array_equal(g1, g2)


def test_patch_stage_harness_random_inputs(orc, mbavo, gpu_ctx):
    """test_compute_patch_cost_gradient_hessian (:897-1011) inputs: random r,J, huber 0.1, inv 1; and the
    cost-only mode leaves slots 1.. untouched (A12)."""
    import torch
    L, O = mbavo.load(), orc.lib()
    rng = np.random.default_rng(5)
    F, K, P, k, E = 5, 145, 8, 4, 325
    res, jac = rng.uniform(-1, 1, F * K * P), rng.uniform(-1, 1, F * K * P * 24)
    o_pb = np.zeros(F * K * E)
    O.orc_compute_patch_cost_gradient_hessian(F, K, P, k, orc.dp(res), orc.dp(jac), 0.1, 1.0, orc.dp(o_pb))
    d_pb = torch.full((F * K * E,), 5.0, dtype=torch.float64, device="cuda:0")
    assert L.mbavo_compute_patch_cost_gradient_hessian(F, K, P, k, _t(res).data_ptr(), _t(jac).data_ptr(), 0.1, 1.0, d_pb.data_ptr()) == 0
    g = d_pb.cpu().numpy()
    assert np.abs(g.reshape(-1, E)[:, 0] - o_pb.reshape(-1, E)[:, 0]).max() < 1e-8
    assert np.abs(g - o_pb).max() < 1e-6      # the harness' bounds
    assert _rel(g, o_pb) < RTOL
    d_pb.fill_(5.0)
    assert L.mbavo_compute_patch_cost_gradient_hessian(F, K, P, k, _t(res).data_ptr(), None, 0.1, 1.0, d_pb.data_ptr()) == 0
    g = d_pb.cpu().numpy().reshape(-1, E)
    assert (g[:, 1:] == 5.0).all() and np.abs(g[:, 0] - o_pb.reshape(-1, E)[:, 0]).max() < 1e-12
    # non-power-of-two patch size: plain sum (the reference's reduce() drops elements there, A10)
    P2 = 5
    res2, jac2 = res[:F * K * P2], jac[:F * K * P2 * 24]
    O.orc_compute_patch_cost_gradient_hessian(F, K, P2, k, orc.dp(res2), orc.dp(jac2), 0.1, 1.0, orc.dp(o_pb))
    assert L.mbavo_compute_patch_cost_gradient_hessian(F, K, P2, k, _t(res2).data_ptr(), _t(jac2).data_ptr(), 0.1, 1.0, d_pb.data_ptr()) == 0
    assert _rel(d_pb.cpu().numpy(), o_pb) < RTOL


def test_input_producers_on_device(orc, mbavo, gpu_ctx):
    """pyramid level + gradient kernels: byte / float exact against the oracle (and thus the reference vectors)."""
    import torch
    L = mbavo.load()
    rng = np.random.default_rng(4)
    for (H, W) in [(480, 640), (50, 66), (61, 35)]:
        src = rng.integers(0, 256, (H, W), dtype=np.uint8)
        d_src = _t(src)
        d_dst = torch.zeros((H // 2) * (W // 2), dtype=torch.uint8, device="cuda:0")
        d_g = torch.zeros(H * W * 2, dtype=torch.float32, device="cuda:0")
        assert L.mbavo_pyramid_down_u8(d_src.data_ptr(), H, W, d_dst.data_ptr(), None) == 0
        assert L.mbavo_image_gradients_u8(d_src.data_ptr(), H, W, d_g.data_ptr(), None) == 0
        torch.cuda.synchronize()
        o_dst = np.zeros((H // 2, W // 2), np.uint8)
        o_g = np.zeros((H, W, 2), np.float32)
        orc.lib().orc_pyramid_down_u8(orc.u8p(src), H, W, orc.u8p(o_dst))
        orc.lib().orc_image_gradients_u8(orc.u8p(src), H, W, orc.fp(o_g), None)
        assert np.array_equal(d_dst.cpu().numpy().reshape(H // 2, W // 2), o_dst)
        assert np.array_equal(d_g.cpu().numpy().reshape(H, W, 2), o_g)


def test_synthetic_blur_generator_on_device(orc, mbavo, gpu_ctx):
    """synthesize_motion_blurred_img (generate_synthetic_data.cpp:182-214) on the device == the oracle's, byte for
    byte: the blurred frames of the tracking tests can be produced by either."""
    import torch
    L = mbavo.load()
    H, W = 120, 160
    ref = synth.texture_image(H, W, seed=9, octaves=(32, 16, 8, 4))
    intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0])
    for k, N in ((4, 6), (2, 4)):
        kt, kR = synth.harness_spline(0.02, 0.3, N)
        kt, kR = np.ascontiguousarray(kt.ravel()), np.ascontiguousarray(kR.ravel())
        o = np.zeros((H, W), np.uint8)
        orc.lib().orc_synthesize_blur(orc.u8p(ref), H, W, 7.5, orc.dp(intr), k, 0.0, 0.5, orc.dp(kt), orc.dp(kR), 0.75, 0.1, 16, orc.u8p(o))
        d_ref = _t(ref)
        d_out = torch.zeros(H * W, dtype=torch.uint8, device="cuda:0")
        assert L.mbavo_synthesize_blur(d_ref.data_ptr(), H, W, 7.5, mbavo.capi.dp(intr), k, 0.0, 0.5, mbavo.capi.dp(kt),
                                       mbavo.capi.dp(kR), N, 0.75, 0.1, 16, d_out.data_ptr(), None) == 0
        g = d_out.cpu().numpy().reshape(H, W)
        assert np.array_equal(g, o)
        assert g.std() > 5 and np.abs(g.astype(int) - ref.astype(int)).max() > 0   # non-trivial image, actually warped


@pytest.mark.parametrize("k", [2, 4])
def test_pose_chain_over_the_whole_range_of_rotations(orc, mbavo, gpu_ctx, k):
    """The device pose chain takes its square roots, reciprocals, atan, sin and cos from short forms written for a rotation's
    half-angle (se3_math.h fastm, round 3) instead of the runtime's general ones; the oracle runs glibc's.  Random knot
    rotations whose RELATIVE angles cover every branch of those forms -- 1e-9 rad (next to the series branches of the
    quaternion log / exp) up to 3.1 rad (the atan's reduction ranges, w of either sign) -- at random blur-sample times:
    poses and both pose-to-knot Jacobians within 1e-12 of the oracle's (observed 1e-15: the forms are good to 1-2 ulp)."""
    import torch
    L, O = mbavo.load(), orc.lib()
    rng = np.random.default_rng(40 + k)
    worst = 0.0
    for angle in (1e-9, 3e-6, 1e-3, 0.05, 0.4, 0.8, 1.1, 1.6, 2.2, 2.9, 3.1):
        F, S, N = 3, 16, 3 + k
        q = np.zeros((N, 4))
        q[0] = rng.normal(size=4)
        q[0] /= np.linalg.norm(q[0])
        for i in range(1, N):  # knot i = knot i-1 * (a rotation by `angle` about a random axis)
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            a = angle * rng.uniform(0.7, 1.0)
            d = np.r_[np.sin(a / 2) * ax, np.cos(a / 2)]
            x1, y1, z1, w1 = q[i - 1]
            x2, y2, z2, w2 = d
            q[i] = [w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                    w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]
        kt = rng.normal(size=(N, 3))
        cap = np.ascontiguousarray(0.2 + 0.5 * np.arange(F) + rng.uniform(0, 0.2, F))
        exp = np.full(F, 0.15)
        nJt, nJR = 9 * k, 12 * k
        o_p, o_Jt, o_JR = np.zeros(F * S * 7), np.zeros(F * S * nJt), np.zeros(F * S * nJR)
        O.orc_compute_virtual_camera_poses(S, F, orc.dp(cap), orc.dp(exp), k, 0.0, 0.5, orc.dp(kt.ravel().copy()),
                                           orc.dp(q.ravel().copy()), orc.dp(o_p), orc.dp(o_Jt), orc.dp(o_JR), None)
        d_p = torch.zeros(F * S * 7, dtype=torch.float64, device="cuda:0")
        d_Jt = torch.zeros(F * S * nJt, dtype=torch.float64, device="cuda:0")
        d_JR = torch.zeros(F * S * nJR, dtype=torch.float64, device="cuda:0")
        assert L.mbavo_compute_virtual_camera_poses(S, F, _t(cap).data_ptr(), _t(exp).data_ptr(), k, 0.0, 0.5, _t(kt.ravel()).data_ptr(),
                                                    _t(q.ravel()).data_ptr(), d_p.data_ptr(), d_Jt.data_ptr(), d_JR.data_ptr()) == 0
        assert np.isfinite(o_JR).all()
        for got, want in ((d_p, o_p), (d_JR, o_JR)):
            r = _rel(got.cpu().numpy(), want)
            worst = max(worst, r)
            assert r < 1e-12, (angle, r)
    assert worst < 1e-12
