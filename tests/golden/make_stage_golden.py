"""Generates tests/golden/ref_stage_vectors.npz by EXECUTING THE REFERENCE's own per-sample code over whole (small)
problems: oracle/_ref's ref_compute_virtual_camera_poses (spline functors), ref_compute_local_patches_xy (Vector3d /
Quaterniond classes) and ref_compute_pixel_jacobian_residual
(compute_pixel_intensity<double> + Core::MatrixMatrixMultiply inside the restated kernel geometry of
compute_hessian_gradients_cost.cu:51-153).  Runs only in the build container (needs /root/reference); the file holds
inputs and outputs, no reference source.  tests/test_oracle_golden.py requires the oracle's stages to reproduce it
bit for bit.

    python tests/golden/make_stage_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import binding as B  # noqa: E402
import scenes  # noqa: E402

B.build()
out = {}
CASES = {"k4": dict(H=60, W=80, S=8, F=2, k=4, P=8, K=30, seed=11),
         "k2": dict(H=60, W=80, S=4, F=1, k=2, P=5, K=40, seed=12, kp="border"),
         "k4_S3": dict(H=30, W=40, S=3, F=1, k=4, P=1, kp="dense", margin=0, seed=13)}
for name, kw in CASES.items():
    sc = scenes.Scene(**kw)
    args = dict(S=sc.S, F=sc.F, K=sc.K, P=sc.P, k=sc.k, N=sc.N, H=sc.H, W=sc.W, ref_img=sc.ref, ref_dIxy=sc.grad,
                cur_imgs=sc.cur, kp_xy=sc.kp_xy, kp_z=sc.kp_z, pattern=sc.pattern, intr=sc.intr, cap=sc.cap, exp_t=sc.exp,
                t0=sc.t0, dt=sc.dt, knots_t=sc.knots_t, knots_R=sc.knots_R, huber_a=sc.huber)
    r = B.stages_with_reference(args)
    out[name + "_kw"] = np.array(repr(kw))
    for key in ("ref_img", "kp_xy", "kp_z", "pattern", "intr", "cap", "exp_t", "knots_t", "knots_R"):
        out["%s_in_%s" % (name, key)] = np.asarray(args[key])
    out[name + "_in_cur"] = np.stack(sc.cur)
    out[name + "_in_scalars"] = np.array([sc.S, sc.F, sc.K, sc.P, sc.k, sc.N, sc.H, sc.W, sc.t0, sc.dt, sc.huber])
    for key in ("poses", "J_t", "J_R", "centres", "residuals", "jacobians", "frame_blocks"):
        out["%s_out_%s" % (name, key)] = r[key]
    print(name, "pixels", sc.F * sc.K * sc.P, "valid residuals", int(np.count_nonzero(r["residuals"])), "cost", r["cost"])
np.savez_compressed(os.path.join(HERE, "ref_stage_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "ref_stage_vectors.npz"), os.path.getsize(os.path.join(HERE, "ref_stage_vectors.npz")), "bytes")
