"""tests/golden/ref_block_vectors.npz (packed_blocks, merge_solve, lm_trace; generator
tests/golden/make_block_golden.py, inputs executed by the reference's own per-sample code and LM / trust-region classes).
CPU part: the oracle reproduces the fixture and the product's HOST logic (merge, solvers, LM classes -- no GPU needed)
agrees with it.  -m gpu part: the reference-named launchers and the tracker on the HIP path against the same arrays."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(HERE, "golden", "ref_block_vectors.npz"))


@pytest.fixture(scope="module")
def stage():
    return np.load(os.path.join(HERE, "golden", "ref_stage_vectors.npz"))


CASES = [(n, a) for n in ("k4", "k2") for a in (0.1, 10.0, 1e32)]


@pytest.mark.parametrize("name,a", CASES)
def test_oracle_reproduces_packed_blocks(orc, vec, stage, name, a):
    """oracle stages 4 / 5 on the reference-executed rows == the committed blocks, bit for bit."""
    L = orc.lib()
    S, F, K, P, k = [int(v) for v in vec[name + "_in_scalars"][:5]]
    inv = float(vec[name + "_in_scalars"][6])
    E = orc.packed_len(k)
    res, jac = np.ascontiguousarray(stage[name + "_out_residuals"]), np.ascontiguousarray(stage[name + "_out_jacobians"])
    tag = "%s_a%g" % (name, a)
    pb = np.zeros(F * K * E)
    L.orc_compute_patch_cost_gradient_hessian(F, K, P, k, orc.dp(res), orc.dp(jac), a, inv, orc.dp(pb))
    assert np.array_equal(pb.reshape(F * K, E), vec[tag + "_patch_blocks"])
    fb, fbm = np.zeros(F * E), np.zeros(F * E)
    mask = np.ascontiguousarray(vec[name + "_mask"])
    L.orc_compute_frame_cost_gradient_hessian(F, K, k, orc.dp(pb), 1, None, orc.dp(fb))
    L.orc_compute_frame_cost_gradient_hessian(F, K, k, orc.dp(pb), 1, orc.u8p(mask), orc.dp(fbm))
    assert np.array_equal(fb.reshape(F, E), vec[tag + "_frame_blocks"])
    assert np.array_equal(fbm.reshape(F, E), vec[tag + "_frame_blocks_masked"])


def _tree_reduce(v):
    """reduce() of ba_tracker/reduction.h:13-55 over the last axis (power-of-two length): buffer[t] += buffer[t + s], s halving."""
    v = np.array(v, dtype=np.float64)
    n = v.shape[-1]
    if n & (n - 1):  # reduce() drops elements of a non-power-of-two buffer; the defined semantics adds all, in order
        acc = np.zeros(v.shape[:-1])
        for i in range(n):
            acc = acc + v[..., i]
        return acc
    while n > 1:
        n //= 2
        v = v[..., :n] + v[..., n:2 * n]
    return v[..., 0]


def kernel_source_blocks(res, jac, F, K, P, k, a, inv, mask=None):
    """The two reduction kernels AS THEIR SOURCE IS WRITTEN (compute_hessian_gradients_cost.cu:165-239 and :247-283), in numpy:
    Huber with the kernel's own fp32 islands -- sqrt_drho_dx = sqrtf(a / (sqrtf(x) + 1e-8)), rho = 2 a sqrtf(x) - a^2 -- the
    weighted row, every product row[i] * row[j] reduced over the patch's pixels by reduce()'s tree and scaled by
    inv_num_residuals; then per frame and entry 256 threads' strided partial sums (flagged keypoints skipped) and the tree.
    An independent restatement from the oracle's C (the harness' formula lacks the + 1e-8 and is held
    to 1e-8 / 1e-6; this one is held to the last bit)."""
    m = 6 * k
    nd = m + 1
    r = res.reshape(F * K, P)
    J = jac.reshape(F * K, P, m)
    x = 0.5 * r * r
    sx = np.sqrt(x.astype(np.float32)).astype(np.float64)                                  # sqrtf(x)
    hub = x > a * a
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        w = np.where(hub, np.sqrt((a / (sx + 1e-8)).astype(np.float32)).astype(np.float64), 1.0)  # sqrtf(a / (sqrtf(x) + 1e-8))
        rho = np.where(hub, 2 * a * sx - a * a, x)
    row = np.concatenate([(w * r)[..., None], w[..., None] * J], axis=2)                   # [patch][pixel][nd]
    E = nd * (nd + 1) // 2
    pb = np.zeros((F * K, E))
    e = 0
    for i in range(nd):
        for j in range(i, nd):
            pb[:, e] = _tree_reduce(row[:, :, i] * row[:, :, j]) * inv
            e += 1
    pb[:, 0] = _tree_reduce(rho) * inv
    fb = np.zeros((F, E))
    pbf = pb.reshape(F, K, E)
    for f in range(F):
        part = np.zeros((256, E))
        for i in range(K):  # thread i % 256 adds its keypoints in ascending order
            if mask is not None and mask[i] == 1:
                continue
            part[i % 256] += pbf[f, i]
        fb[f] = _tree_reduce(part.T)
    return pb, fb


@pytest.mark.parametrize("name,a", CASES)
def test_blocks_equal_the_kernel_source_restated_in_numpy(vec, stage, name, a):
    """a6 / a7 pinned tighter than the harness' tolerances: the committed patch and frame blocks (oracle outputs on the
    reference-executed rows) equal a second, independent restatement of the kernels' SOURCE -- numpy, written from the .cu text
    with its sqrtf islands, its + 1e-8 and reduce()'s tree -- to the last bit, with and without outlier flags."""
    S, F, K, P, k = [int(v) for v in vec[name + "_in_scalars"][:5]]  # (k4: 8-pixel patches, reduce()'s tree; k2: 5 pixels, A10)
    inv = float(vec[name + "_in_scalars"][6])
    res, jac = np.ascontiguousarray(stage[name + "_out_residuals"]), np.ascontiguousarray(stage[name + "_out_jacobians"])
    tag = "%s_a%g" % (name, a)
    pb, fb = kernel_source_blocks(res, jac, F, K, P, k, a, inv)
    assert np.array_equal(pb, vec[tag + "_patch_blocks"]), np.abs(pb - vec[tag + "_patch_blocks"]).max()
    assert np.array_equal(fb, vec[tag + "_frame_blocks"]), np.abs(fb - vec[tag + "_frame_blocks"]).max()
    _, fbm = kernel_source_blocks(res, jac, F, K, P, k, a, inv, mask=vec[name + "_mask"])
    assert np.array_equal(fbm, vec[tag + "_frame_blocks_masked"])
    if a < 1e30:
        assert (0.5 * res * res > a * a).any(), "no pixel in the Huber branch: the case tests nothing of it"


def test_merge_equals_the_source_restated_in_numpy(vec):
    """a8 the same way: merge_hessian_gradient_cost.cpp:39-86 written out in numpy from its text (gradient halves at
    3 * start and 3 * (N + start), the packed upper triangle walked row by row with the two offsets, both triangles of H, the
    frames in ascending order) against the committed merged system, to the last bit."""
    fb3, start = vec["merge_frame_blocks"], vec["merge_start"]
    k, N = 4, 6
    nd, W = 6 * k + 1, 6 * N
    H, g, cost = np.zeros((W, W)), np.zeros(W), 0.0
    for i in range(fb3.shape[0]):
        st, blk = int(start[i]), fb3[i]
        cost += blk[0]
        g[3 * st:3 * st + 3 * k] += blk[1:1 + 3 * k]
        g[3 * (N + st):3 * (N + st) + 3 * k] += blk[1 + 3 * k:1 + 6 * k]
        off0, off1 = 3 * st, 3 * (N + st)
        e = nd
        for j in range(nd - 1):
            r = j + (off0 if j < 3 * k else off1 - 3 * k)
            for c_ in range(j, nd - 1):
                c = c_ + (off0 if c_ < 3 * k else off1 - 3 * k)
                H[r, c] += blk[e]
                if c != r:
                    H[c, r] += blk[e]
                e += 1
    assert np.array_equal(H.T.ravel(), vec["merge_H_colmajor"]) and np.array_equal(g, vec["merge_g"]) and cost == vec["merge_cost"][0]


def test_host_merge_and_solvers_match_fixture(orc, mbavo, vec):
    """merge_hessian_gradient_cost (product host code and oracle) == the fixture exactly; the product's SVD / LDLT
    solvers on the damped system within rounding x cond(H) = 6e9 (1e-5 relative), residual ||Hx + g|| <= 1e-9 ||g||;
    minimum-norm solution on the rank-deficient 7-knot system: zero step on the untouched knot."""
    lib = mbavo.load()
    fb3, start = np.ascontiguousarray(vec["merge_frame_blocks"]), np.ascontiguousarray(vec["merge_start"])
    n = 36
    for fn in (lib.mbavo_merge_host, orc.lib().orc_merge_hessian_gradient_cost):
        cost, H, g = np.zeros(1), np.zeros(n * n), np.zeros(n)
        fn(3, 4, mbavo.capi.dp(fb3), mbavo.capi.ip(start), 6, mbavo.capi.dp(cost), mbavo.capi.dp(H), mbavo.capi.dp(g))
        assert np.array_equal(H, vec["merge_H_colmajor"]) and np.array_equal(g, vec["merge_g"]) and cost[0] == vec["merge_cost"][0]
    Hd, g, xg = np.ascontiguousarray(vec["solve_H_damped_colmajor"]), np.ascontiguousarray(vec["merge_g"]), vec["solve_x_damped"]
    Hm = Hd.reshape(n, n).T
    for solver in (0, 1):
        x = np.zeros(n)
        assert lib.mbavo_solve_normal_equation(mbavo.capi.dp(Hd), mbavo.capi.dp(g), n, solver, mbavo.capi.dp(x)) == 0
        assert np.linalg.norm(x - xg) <= 1e-5 * np.linalg.norm(xg)
        assert np.linalg.norm(Hm @ x + g) <= 1e-9 * np.linalg.norm(g)
        xo = np.zeros(n)
        orc.lib().orc_solve_normal_equation(orc.dp(Hd), orc.dp(g), n, solver, orc.dp(xo))
        assert np.linalg.norm(xo - xg) <= 1e-5 * np.linalg.norm(xg)
    H7, g7, x7 = np.ascontiguousarray(vec["solve_H7_colmajor"]), np.ascontiguousarray(vec["solve_g7"]), vec["solve_x7_minnorm"]
    x = np.zeros(42)
    assert lib.mbavo_solve_normal_equation(mbavo.capi.dp(H7), mbavo.capi.dp(g7), 42, 0, mbavo.capi.dp(x)) == 0
    assert np.linalg.norm(x - x7) <= 1e-5 * np.linalg.norm(x7)
    untouched = np.r_[18:21, 39:42]
    assert np.abs(x[untouched]).max() <= 1e-12 * np.abs(x).max()


@pytest.mark.parametrize("name", ["lm_k4", "lm_k2"])
def test_oracle_loop_reproduces_lm_trace(orc, vec, name):
    """The oracle's own C loop (orc_optimize_trajectory) against the trace recorded with the REFERENCE's compiled
    LM strategy / step evaluator driving the loop: identical (level, iteration, kind, outliers), costs 1e-9."""
    import tracking
    sc = tracking.make_tracking_scene(orc, **eval(str(vec[name + "_kw"])))
    ro = tracking.run_oracle_tracker(orc, sc, tracking.OPTS)
    tr = vec[name + "_trace"]
    assert len(ro["trace"]) == len(tr)
    for a, b in zip(ro["trace"], tr):
        assert tuple(a[:4]) == tuple(int(v) for v in b[:4])
        assert a[4] == pytest.approx(b[4], rel=1e-6) and a[5] == pytest.approx(b[5], rel=1e-9) and a[6] == pytest.approx(b[6], rel=1e-9, abs=1e-12)


def test_product_lm_classes_replay_fixture(mbavo, vec):
    """LevenbergMarquardtStrategy / TrustRegionStepEvaluator of the product (host C++, through the C ABI) replayed over
    the recorded decisions: the radius after every record and the step quality of every evaluated candidate."""
    lib = mbavo.load()
    for name in ("lm_k4", "lm_k2"):
        tr = vec[name + "_trace"]
        lm, ev = lib.mbavo_lm_new(), lib.mbavo_tr_new(5)
        for lv, it, kind, nout, radius, ec, cc, model, q in tr:
            kind = int(kind)
            if kind == 0:
                lib.mbavo_lm_reset(lm)
                lib.mbavo_tr_reset(ev, ec)
            elif kind == 3:
                lib.mbavo_lm_step_rejected(lm)
            else:
                assert lib.mbavo_tr_step_quality(ev, cc, model) == pytest.approx(q, rel=1e-12)
                if kind == 1:
                    lib.mbavo_lm_step_accepted(lm, q)
                    lib.mbavo_tr_step_accepted(ev, ec, model)
                else:
                    lib.mbavo_lm_step_rejected(lm)
            assert lib.mbavo_lm_get_radius(lm) == pytest.approx(radius, rel=1e-12)
        lib.mbavo_lm_delete(lm)
        lib.mbavo_tr_delete(ev)


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name,a", CASES)
def test_gpu_launchers_match_packed_blocks(mbavo, gpu_ctx, vec, stage, name, a):
    """compute_patch_cost_gradient_hessian / compute_frame_cost_gradient_hessian (the reference-named launchers on the
    HIP path) on the reference-executed rows against the committed blocks: 1e-12 relative (patch blocks: the kernel's
    in-lane sums over the P pixels; frame blocks: its fixed-order tree over the K patches)."""
    import torch
    S, F, K, P, k = [int(v) for v in vec[name + "_in_scalars"][:5]]
    inv = float(vec[name + "_in_scalars"][6])
    E = mbavo.load().mbavo_packed_len(k)
    tag = "%s_a%g" % (name, a)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    res, jac, mask = t(stage[name + "_out_residuals"]), t(stage[name + "_out_jacobians"]), t(vec[name + "_mask"])
    pb = torch.zeros(F * K * E, dtype=torch.float64, device="cuda:0")
    lib = gpu_ctx.lib
    assert lib.mbavo_compute_patch_cost_gradient_hessian(F, K, P, k, res.data_ptr(), jac.data_ptr(), a, inv, pb.data_ptr()) == 0
    want = vec[tag + "_patch_blocks"]
    got = pb.cpu().numpy().reshape(F * K, E)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    for m, key in ((None, "_frame_blocks"), (mask, "_frame_blocks_masked")):
        fb = torch.zeros(F * E, dtype=torch.float64, device="cuda:0")
        assert lib.mbavo_compute_frame_cost_gradient_hessian(F, K, k, pb.data_ptr(), 1, m.data_ptr() if m is not None else None,
                                                             fb.data_ptr()) == 0
        w = vec[tag + key]
        assert np.abs(fb.cpu().numpy().reshape(F, E) - w).max() <= 1e-12 * np.abs(w).max()


@pytest.mark.gpu
def test_gpu_merge_device_matches_fixture(mbavo, gpu_ctx, vec):
    """mbavo_merge_device (merge_hessian_gradient_cost on the device) on the fixture's frame blocks: identical bits."""
    import torch
    fb3 = torch.from_numpy(np.ascontiguousarray(vec["merge_frame_blocks"])).to("cuda:0")
    start = np.ascontiguousarray(vec["merge_start"])
    p = mbavo.capi.Problem()
    p.F, p.N = 3, 6
    p.h_start_idx = start.ctypes.data_as(C.POINTER(C.c_int))
    arr = (mbavo.capi.Problem * 1)(p)
    n = gpu_ctx.lib.mbavo_system_len(6)
    sysd = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    assert gpu_ctx.lib.mbavo_merge_device(gpu_ctx.handle, 1, arr, 4, fb3.data_ptr(), sysd.data_ptr()) == 0
    torch.cuda.synchronize()
    s = sysd.cpu().numpy()
    assert s[0] == vec["merge_cost"][0] and np.array_equal(s[1:37], vec["merge_g"]) and np.array_equal(s[37:], vec["merge_H_colmajor"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lm_k4", "lm_k2"])
def test_gpu_tracker_reproduces_lm_trace(orc, mbavo, gpu_ctx, vec, name):
    """mbavo_optimize_trajectory on the HIP path against the recorded trace: identical (level, iteration, kind,
    outliers); costs 1e-6 relative (solver rounding x cond(H), see tests/test_gpu_tracker.py)."""
    import tracking
    sc = tracking.make_tracking_scene(orc, **eval(str(vec[name + "_kw"])))
    rg = tracking.run_gpu_tracker(mbavo, gpu_ctx, sc, tracking.OPTS)
    tr = vec[name + "_trace"]
    assert len(rg["trace"]) == len(tr)
    for a, b in zip(rg["trace"], tr):
        assert tuple(a[:4]) == tuple(int(v) for v in b[:4])
        assert a[5] == pytest.approx(b[5], rel=1e-6) and a[6] == pytest.approx(b[6], rel=1e-6, abs=1e-12)
