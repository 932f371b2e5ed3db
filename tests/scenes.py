"""Shared scene construction for tests: numpy problems for the oracle and their
device-resident twins for the HIP library (torch is only device-memory plumbing)."""
import ctypes as C

import numpy as np

import mba_vo_amd as M
from mba_vo_amd import synth


class Scene:
    """One alignment problem (all numpy, host)."""

    def __init__(self, H=480, W=640, S=8, F=1, k=4, N=None, P=8, K=145, seed=0, image="noise", cur="warp",
                 kp="harness", huber=10.0, intr=None, trans_scale=0.004, rot_scale=0.05, t0=0.0, dt=0.5,
                 exp=0.1, cap0=0.25, outlier_frac=0.0, z_range=(5.0, 10.0), margin=24, pattern=None):
        rng = np.random.default_rng(seed)
        self.H, self.W, self.S, self.F, self.k, self.P = H, W, S, F, k, P
        self.t0, self.dt, self.huber = t0, dt, huber
        self.intr = np.array(intr if intr is not None else [W / 2.0, W / 2.0, W / 2.0, H / 2.0], np.float64)
        if image == "ramp":
            self.ref = synth.ramp_image(H, W)
        elif image == "shapes":
            self.ref = synth.shapes_image(H, W)
        else:
            self.ref = synth.noise_image(H, W, seed=seed + 1)
        self.grad = synth.image_gradients(self.ref)
        self.cap = np.ascontiguousarray(cap0 + dt * np.arange(F), np.float64)
        self.exp = np.full(F, exp, np.float64)
        n_knots = N if N is not None else F + k - 1 + (0 if k == 2 else 0)
        n_knots = max(n_knots, int((self.cap[-1] + exp - t0) / dt) + k)
        self.N = n_knots
        kt, kR = synth.harness_spline(trans_scale, rot_scale, n_knots)
        kt = kt + rng.normal(0, 1e-3, kt.shape)
        self.knots_t = np.ascontiguousarray(kt.ravel())
        self.knots_R = np.ascontiguousarray(kR.ravel())
        self.start_idx = np.array([synth.segment_start_index(c, t0, dt) for c in self.cap], np.int32)
        if pattern is not None:
            self.pattern = np.ascontiguousarray(pattern, np.int32)
        elif P == 8:
            self.pattern = synth.PATTERN8.copy()
        elif P == 1:
            self.pattern = np.zeros(2, np.int32)
        else:
            self.pattern = np.ascontiguousarray(rng.integers(-3, 4, 2 * P), np.int32)
        if kp == "dense":
            self.kp_xy, self.kp_z = synth.dense_keypoints(H, W, margin=margin, z_lo=z_range[0], z_hi=z_range[1], seed=seed + 2)
        elif kp == "border":  # keypoints everywhere incl. the image border: exercises out-of-bounds handling
            self.kp_xy = np.ascontiguousarray(np.stack([rng.uniform(-4, W + 4, K), rng.uniform(-4, H + 4, K)], 1))
            self.kp_z = rng.uniform(z_range[0], z_range[1], K)
        else:
            self.kp_xy = np.ascontiguousarray(np.stack([rng.integers(margin, W - margin, K),
                                                        rng.integers(margin, H - margin, K)], 1).astype(np.float64))
            self.kp_z = rng.uniform(z_range[0], z_range[1], K)
        self.K = self.kp_xy.shape[0]
        # current images: the keyframe itself, or a perturbed copy so residuals are non-trivial
        self.cur = []
        for f in range(F):
            if cur == "same":
                self.cur.append(self.ref.copy())
            else:
                sh = np.roll(self.ref, (f + 1, -(f + 2)), (0, 1)).astype(np.int32)
                noise = rng.integers(-6, 7, self.ref.shape)
                self.cur.append(np.ascontiguousarray(np.clip(sh + noise, 0, 255).astype(np.uint8)))
        self.outlier = None
        self.num_bad = 0
        if outlier_frac > 0:
            self.outlier = (rng.random(self.K) < outlier_frac).astype(np.uint8)
            self.num_bad = int(self.outlier.sum())

    @property
    def E(self):
        return synth.packed_len(self.k)

    def oracle_problem(self, orc):
        return orc.make_problem(self.S, self.F, self.K, self.P, self.k, self.N, self.H, self.W, self.ref, self.grad,
                                self.cur, self.kp_xy, self.kp_z, self.pattern, self.intr, self.cap, self.exp,
                                self.t0, self.dt, self.knots_t, self.knots_R, self.start_idx, self.huber,
                                outlier=self.outlier, num_bad=self.num_bad)


class DeviceScene:
    """Device-resident copy of a Scene + its mbavo_problem."""

    def __init__(self, sc, vec2d=False, packed=False):
        import torch
        dev = "cuda:0"
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.sc = sc
        self.ref = t(sc.ref)
        # packed: the keyframe as one word per pixel (mbavo_problem.grad_fp16 = 2) instead of the float gradient image
        self.packed = packed
        self.grad = t(synth.pack_keyframe(sc.ref).view(np.int32)) if packed else t(sc.grad)
        self.cur = [t(c) for c in sc.cur]
        self.cur_ptrs = torch.tensor([c.data_ptr() for c in self.cur], dtype=torch.int64, device=dev)
        if vec2d:  # Core::Vector2d array: {int nDim; pad; double x; double y} = 3 doubles
            v = np.zeros((sc.K, 3), np.float64)
            v.view(np.int32)[:, 0] = 2
            v[:, 1:] = sc.kp_xy
            self.kp_vec2d = t(v)
            self.kp_xy_ptr = self.kp_vec2d.data_ptr() + 8
            self.kp_stride = 3
        else:
            self.kp_xy = t(sc.kp_xy)
            self.kp_xy_ptr = self.kp_xy.data_ptr()
            self.kp_stride = 2
        self.kp_z = t(sc.kp_z)
        self.pattern = t(sc.pattern)
        self.outlier = t(sc.outlier) if sc.outlier is not None else None
        self.cap = t(sc.cap)
        self.exp = t(sc.exp)
        self.knots_t = t(sc.knots_t)
        self.knots_R = t(sc.knots_R)
        self.start_idx = np.ascontiguousarray(sc.start_idx, np.int32)
        torch.cuda.synchronize()

    def problem(self):
        sc = self.sc
        p = M.capi.Problem()
        p.S, p.F, p.K, p.P, p.N, p.H, p.W = sc.S, sc.F, sc.K, sc.P, sc.N, sc.H, sc.W
        p.d_ref_img = self.ref.data_ptr()
        p.d_ref_dIxy = self.grad.data_ptr()
        p.d_cur_imgs = self.cur_ptrs.data_ptr()
        p.d_kp_xy = self.kp_xy_ptr
        p.kp_stride = self.kp_stride
        p.d_kp_z = self.kp_z.data_ptr()
        p.d_pattern = self.pattern.data_ptr()
        p.d_outlier = self.outlier.data_ptr() if self.outlier is not None else None
        p.num_bad = sc.num_bad
        for i in range(4):
            p.intrinsics[i] = float(sc.intr[i])
        p.d_cap_time = self.cap.data_ptr()
        p.d_exp_time = self.exp.data_ptr()
        p.t0, p.dt = sc.t0, sc.dt
        p.d_knots_t = self.knots_t.data_ptr()
        p.d_knots_R = self.knots_R.data_ptr()
        p.h_start_idx = self.start_idx.ctypes.data_as(C.POINTER(C.c_int))
        p.huber_a = sc.huber
        p.grad_fp16 = 2 if self.packed else 0
        return p


def gpu_eval(ctx, dsc, with_hessian=True, patch_blocks=None):
    """mbavo_eval -> dict(cost, H, g)."""
    sc = dsc.sc
    n = 6 * sc.N
    cost = np.zeros(1)
    H = np.zeros(n * n) if with_hessian else None
    g = np.zeros(n) if with_hessian else None
    p = dsc.problem()
    rc = ctx.lib.mbavo_eval(ctx.handle, C.byref(p), sc.k, M.capi.dp(cost), M.capi.dp(H), M.capi.dp(g),
                            patch_blocks.data_ptr() if patch_blocks is not None else None)
    M.capi.check(rc, "mbavo_eval")
    return dict(cost=float(cost[0]), H=None if H is None else H.reshape(n, n).T.copy(), g=g)


def gpu_eval_batch(ctx, dscs, k, with_hessian=True):
    """mbavo_eval_batch -> (frame_blocks [sumF, E], patch_cost, valid) as numpy."""
    import torch
    B = len(dscs)
    arr = (M.capi.Problem * B)(*[d.problem() for d in dscs])
    E = synth.packed_len(k)
    nbf = sum(d.sc.F for d in dscs)
    npatch = sum(d.sc.F * d.sc.K for d in dscs)
    fb = torch.zeros(nbf * E, dtype=torch.float64, device="cuda:0")
    pc = torch.zeros(max(npatch, 1), dtype=torch.float64, device="cuda:0")
    valid = torch.zeros(nbf, dtype=torch.float64, device="cuda:0")
    rc = ctx.lib.mbavo_eval_batch(ctx.handle, B, arr, k, 1 if with_hessian else 0, fb.data_ptr(), pc.data_ptr(),
                                  valid.data_ptr())
    M.capi.check(rc, "mbavo_eval_batch")
    torch.cuda.synchronize()
    return fb.cpu().numpy().reshape(nbf, E), pc.cpu().numpy()[:npatch], valid.cpu().numpy()
