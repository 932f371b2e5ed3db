"""The two FAST forms of the oracle that bench.py and the full-size GPU tests use -- orc_evaluate_fast (fused OpenMP port,
`cpu_baseline.kind = "port"`, tests/test_gpu_fullsize.py) and evaluate_with_reference (the reference's own per-sample code
from oracle/_ref in keypoint chunks, `cpu_baseline.kind = "reference"`) -- against the plain restatement orc_evaluate that
the golden vectors pin (tests/test_oracle_golden.py).  CPU only."""
import numpy as np
import pytest

import scenes

CASES = {
    "k4_S8_P8_F2": dict(H=120, W=160, S=8, F=2, k=4, P=8, K=90, margin=12),
    "k2_S4_P8": dict(H=120, W=160, S=4, F=1, k=2, P=8, K=90, margin=12),
    "k4_S3_P5_border_outliers": dict(H=120, W=160, S=3, F=2, k=4, P=5, K=120, kp="border", outlier_frac=0.1),
    "k4_S1_dense": dict(H=40, W=56, S=1, F=1, k=4, P=1, kp="dense", margin=0),
    "k4_S16_dense_6knots": dict(H=36, W=48, S=16, F=2, k=4, P=1, kp="dense", margin=1, N=6),
    "k2_huber_small": dict(H=120, W=160, S=8, F=1, k=2, P=8, K=60, margin=12, huber=0.1),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_evaluate_fast_equals_evaluate(orc, name):
    """Frame blocks, cost, H and g of the fused port: 1e-13 relative to the plain restatement (the port sums the pixels of
    a thread's keypoints in another grouping; the per-pixel arithmetic is the same code) on 1 and 3 threads, H/g and
    cost-only."""
    sc = scenes.Scene(**CASES[name])
    p, keep = sc.oracle_problem(orc)
    want = orc.evaluate(p)
    for threads in (1, 3):
        got = orc.evaluate_fast(p, num_threads=threads)
        scale = np.abs(want["frame_blocks"]).max(axis=1, keepdims=True)
        assert (np.abs(got["frame_blocks"] - want["frame_blocks"]) <= 1e-13 * scale).all(), threads
        assert abs(got["cost"] - want["cost"]) <= 1e-13 * abs(want["cost"])
        assert np.abs(got["H"] - want["H"]).max() <= 1e-13 * np.abs(want["H"]).max()
        assert np.abs(got["g"] - want["g"]).max() <= 1e-13 * np.abs(want["g"]).max()
    wc, gc = orc.evaluate(p, with_hessian=False), orc.evaluate_fast(p, num_threads=2, with_hessian=False)
    assert abs(gc["cost"] - wc["cost"]) <= 1e-13 * abs(wc["cost"])


@pytest.mark.parametrize("name", ["k4_S8_P8_F2", "k2_S4_P8", "k4_S1_dense", "k4_S16_dense_6knots"])
@pytest.mark.parametrize("omp", ["1", "0"])
def test_evaluate_with_reference_live(orc, monkeypatch, name, omp):
    """Where oracle/_ref is present (this container; it travels to the GPU box as a built .so): the evaluation assembled
    from the REFERENCE's compiled per-sample code, in keypoint chunks on 1 and 2 threads, against the restatement.  No
    outliers (the chunked driver has no flags).  1e-12: the chunks' frame sums are added in another grouping.
    omp = 1 (round 4, the default): the chunk loop is an OpenMP loop inside oracle/ref_shim.cpp (ref_evaluate_omp); 0: the
    Python thread pool over stages_with_reference."""
    monkeypatch.setenv("MBAVO_REF_OMP", omp)
    if orc.ref() is None or not hasattr(orc.ref(), "ref_compute_pixel_jacobian_residual"):
        pytest.skip("oracle/_ref is not built here")
    sc = scenes.Scene(**CASES[name])
    p, keep = sc.oracle_problem(orc)
    want = orc.evaluate(p)["frame_blocks"]
    a = dict(S=sc.S, F=sc.F, K=sc.K, P=sc.P, k=sc.k, N=sc.N, H=sc.H, W=sc.W, ref_img=sc.ref, ref_dIxy=sc.grad, cur_imgs=sc.cur,
             kp_xy=sc.kp_xy, kp_z=sc.kp_z, pattern=sc.pattern, intr=sc.intr, cap=sc.cap, exp_t=sc.exp, t0=sc.t0, dt=sc.dt,
             knots_t=sc.knots_t, knots_R=sc.knots_R, huber_a=sc.huber)
    for chunk, threads in ((4096, 1), (37, 1), (37, 2)):
        got = orc.evaluate_with_reference(a, chunk=chunk, threads=threads)
        scale = np.abs(want).max(axis=1, keepdims=True)
        assert (np.abs(got - want) <= 1e-12 * scale).all(), (chunk, threads)


@pytest.mark.parametrize("name", ["k4_S8_P8_F2", "k4_S3_P5_border_outliers", "k4_S1_dense", "k4_S16_dense_6knots"])
def test_count_valid_against_the_reference_stage(orc, name):
    """orc_count_valid (the exact valid-pixel counts of the full-size GPU tests): a pixel is valid iff its integer location and
    all S warps are in bounds.  Pinned on the REFERENCE's compiled per-sample stage where oracle/_ref is present -- an invalid
    pixel leaves residual 0 and an all-zero Jacobian row there (compute_pixel_jacobian_residual.cu:69-120), a valid one does not
    -- else on the restatement's stage; the same on 1 and 3 threads; scenes with keypoints on the border have invalid pixels."""
    sc = scenes.Scene(**CASES[name])
    p, keep = sc.oracle_problem(orc)
    v1, v3 = orc.count_valid(p, 1), orc.count_valid(p, 3)
    assert np.array_equal(v1, v3) and (v1 <= sc.K * sc.P).all() and v1.sum() > 0
    if orc.ref() is not None and hasattr(orc.ref(), "ref_compute_pixel_jacobian_residual"):
        a = dict(S=sc.S, F=sc.F, K=sc.K, P=sc.P, k=sc.k, N=sc.N, H=sc.H, W=sc.W, ref_img=sc.ref, ref_dIxy=sc.grad, cur_imgs=sc.cur,
                 kp_xy=sc.kp_xy, kp_z=sc.kp_z, pattern=sc.pattern, intr=sc.intr, cap=sc.cap, exp_t=sc.exp, t0=sc.t0, dt=sc.dt,
                 knots_t=sc.knots_t, knots_R=sc.knots_R, huber_a=sc.huber)
        st = orc.stages_with_reference(a)
        res = np.asarray(st["residuals"]).reshape(sc.F, sc.K * sc.P)
        jac = np.asarray(st["jacobians"]).reshape(sc.F, sc.K * sc.P, 6 * sc.k)
        live = (res != 0) | (jac != 0).any(axis=2)
        assert np.array_equal(live.sum(axis=1).astype(float), v1), (live.sum(axis=1), v1)
    if "border" in name:
        assert (v1 < sc.K * sc.P).all()
