"""BASELINE.json workloads as lists of alignment problems (numpy), plus device upload.

A workload is a list of `Prob` (one per pyramid level or per keyframe pair).  bench.py
times `mbavo_eval_batch` over the whole list (one GN iteration = one H/g evaluation of
every problem in it); tests use the same builders at reduced size.  Pure numpy + torch
plumbing; no oracle code.
"""
import ctypes as C

import numpy as np

from . import capi, synth


class Prob:
    """One alignment problem on the host (numpy arrays; layout of include/mbavo.h:mbavo_problem)."""

    def __init__(self, ref, cur, kp_xy, kp_z, pattern, intr, S, k, N, cap, exp, t0, dt, knots_t, knots_R, huber,
                 grad=None):
        self.ref = np.ascontiguousarray(ref)
        self.grad = synth.image_gradients(self.ref) if grad is None else grad
        self.cur = [np.ascontiguousarray(c) for c in cur]
        self.H, self.W = self.ref.shape
        self.kp_xy, self.kp_z = np.ascontiguousarray(kp_xy), np.ascontiguousarray(kp_z)
        self.pattern = np.ascontiguousarray(pattern, np.int32)
        self.intr = np.ascontiguousarray(intr, np.float64)
        self.S, self.k, self.N, self.F = S, k, N, len(self.cur)
        self.K, self.P = self.kp_xy.shape[0], self.pattern.size // 2
        self.cap, self.exp = np.ascontiguousarray(cap, np.float64), np.ascontiguousarray(exp, np.float64)
        self.t0, self.dt, self.huber = float(t0), float(dt), float(huber)
        self.knots_t, self.knots_R = np.ascontiguousarray(knots_t, np.float64).ravel(), np.ascontiguousarray(knots_R, np.float64).ravel()
        self.start_idx = np.array([synth.segment_start_index(c, t0, dt) for c in self.cap], np.int32)
        self.outlier, self.num_bad = None, 0
        self.grad_fp16 = False  # upload the gradient image as IEEE half pairs (BASELINE configs[4])

    @property
    def pixel_samples(self):
        return self.F * self.K * self.P * self.S


def _current_image(ref, rng, shift=(1, -2), noise=6):
    sh = np.roll(ref, shift, (0, 1)).astype(np.int32)
    return np.ascontiguousarray(np.clip(sh + rng.integers(-noise, noise + 1, ref.shape), 0, 255).astype(np.uint8))


def pyramid_pair(H=480, W=640, levels=4, S=8, k=4, N=4, mode="dense", seed=1, huber=10.0, margin=0,
                 trans_scale=0.004, rot_scale=0.05, frames=1):
    """Config 1/2 of BASELINE.json: one keyframe pair, `levels` pyramid levels, `S` blur samples,
    `N` control poses.  mode 'dense': every pixel a P=1 patch with its own depth;
    'semidense': grid-selected keypoints (30-px cells) with the harness' 8-pixel pattern.
    frames > 1: a joint problem of `frames` blurred frames against the same keyframe on one spline segment (the
    multi-GPU workload: one frame per rank); frame 0 is exactly the frames == 1 problem."""
    rng = np.random.default_rng(seed)
    ref0 = synth.noise_image(H, W, seed=seed)
    refs = synth.pyramid(ref0, levels)
    curs = [synth.pyramid(_current_image(ref0, rng, shift=(1 + f % 3, -2 - f % 2)), levels) for f in range(frames)]
    kt, kR = synth.harness_spline(trans_scale, rot_scale, N)
    t0, dt = 0.0, 0.5
    cap, exp = [0.25 + 0.01 * f for f in range(frames)], [0.1] * frames
    assert all(synth.segment_start_index(c - 0.05, t0, dt) == 0 and synth.segment_start_index(c + 0.05, t0, dt) + k <= N for c in cap)
    probs = []
    for l in range(levels):
        sc = 2 ** l
        Hl, Wl = refs[l].shape
        intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0]) / sc
        if mode == "dense":
            xy, z = synth.dense_keypoints(Hl, Wl, margin=margin, seed=seed + 10 + l)
            pat = np.zeros(2, np.int32)
        else:
            xy, z = synth.semi_dense_keypoints(refs[l], cell=30, thresh=4.0, margin=max(4, 20 // sc), seed=seed + 10 + l)
            pat = synth.PATTERN8
        probs.append(Prob(refs[l], [c[l] for c in curs], xy, z, pat, intr, S, k, N, cap, exp, t0, dt, kt, kR, huber))
    return probs


def pair_batch(B, H=480, W=640, S=8, k=4, N=4, mode="semidense", seed=1, huber=10.0):
    """Config 2/3 of BASELINE.json: B independent keyframe pairs (consecutive frames of one
    synthetic sequence, each with its own knots), level 0 only."""
    rng = np.random.default_rng(seed)
    ref0 = synth.noise_image(H, W, seed=seed)
    grad0 = synth.image_gradients(ref0)
    if mode == "dense":
        xy, z = synth.dense_keypoints(H, W, seed=seed + 10)
        pat = np.zeros(2, np.int32)
    else:
        xy, z = synth.semi_dense_keypoints(ref0, cell=30, thresh=4.0, margin=20, seed=seed + 10)
        pat = synth.PATTERN8
    intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0])
    probs = []
    for b in range(B):
        kt, kR = synth.harness_spline(0.004, 0.05, N)
        kt = kt + rng.normal(0, 2e-3, kt.shape)
        cur = _current_image(ref0, rng, shift=(1 + b % 3, -(1 + b % 2)))
        probs.append(Prob(ref0, [cur], xy, z, pat, intr, S, k, N, [0.25], [0.1], 0.0, 0.5, kt, kR, huber, grad=grad0))
    return probs


class DeviceWorkload:
    """Uploads a list of Prob once; builds the mbavo_problem array (inputs resident in HBM)."""

    def __init__(self, probs, device="cuda:0"):
        import torch
        self.probs = probs
        self.keep = []
        cache = {}

        def up(a):
            key = (a.__array_interface__["data"][0], a.shape, a.dtype.str)
            if key not in cache:
                cache[key] = torch.from_numpy(a).to(device)
            return cache[key]

        B = len(probs)
        self._knots = []
        self.array = (capi.Problem * B)()
        for b, p in enumerate(probs):
            ref = up(p.ref)
            if p.grad_fp16:
                if not hasattr(p, "_grad_half"):
                    p._grad_half = np.ascontiguousarray(p.grad.astype(np.float16))
                grad = up(p._grad_half)
            else:
                grad = up(p.grad)
            curs = [up(c) for c in p.cur]
            cur_ptrs = torch.tensor([c.data_ptr() for c in curs], dtype=torch.int64, device=device)
            xy, z, pat = up(p.kp_xy), up(p.kp_z), up(p.pattern)
            cap, exp, kt, kR = up(p.cap), up(p.exp), up(p.knots_t), up(p.knots_R)
            out = up(p.outlier) if p.outlier is not None else None
            self.keep += [ref, grad, curs, cur_ptrs, xy, z, pat, cap, exp, kt, kR, out]
            q = self.array[b]
            q.S, q.F, q.K, q.P, q.N, q.H, q.W = p.S, p.F, p.K, p.P, p.N, p.H, p.W
            q.d_ref_img, q.d_ref_dIxy, q.d_cur_imgs = ref.data_ptr(), grad.data_ptr(), cur_ptrs.data_ptr()
            q.d_kp_xy, q.kp_stride, q.d_kp_z, q.d_pattern = xy.data_ptr(), 2, z.data_ptr(), pat.data_ptr()
            q.d_outlier, q.num_bad = (out.data_ptr() if out is not None else None), p.num_bad
            for i in range(4):
                q.intrinsics[i] = float(p.intr[i])
            q.d_cap_time, q.d_exp_time, q.t0, q.dt = cap.data_ptr(), exp.data_ptr(), p.t0, p.dt
            q.d_knots_t, q.d_knots_R = kt.data_ptr(), kR.data_ptr()
            q.h_start_idx = p.start_idx.ctypes.data_as(C.POINTER(C.c_int))
            q.huber_a = p.huber
            q.grad_fp16 = 1 if p.grad_fp16 else 0
            self._knots.append((kt, kR))
        self.B = B
        self.k = probs[0].k
        self.E = synth.packed_len(self.k)
        self.nbf = sum(p.F for p in probs)
        self.frame_blocks = torch.zeros(self.nbf * self.E, dtype=torch.float64, device=device)
        self.valid = torch.zeros(self.nbf, dtype=torch.float64, device=device)
        torch.cuda.synchronize()

    def keep_knots(self, b):
        """(knots_t, knots_R) device tensors of problem b (updated in place by mbavo_lm_batch)."""
        return self._knots[b]

    def step(self, ctx, with_hessian=True, out=None):
        """One GN-iteration evaluation of every problem (asynchronous on the context's stream); `out` replaces the
        default output tensor (double buffering under an asynchronous all-reduce)."""
        fb = self.frame_blocks if out is None else out
        rc = ctx.lib.mbavo_eval_batch(ctx.handle, self.B, self.array, self.k, 1 if with_hessian else 0,
                                      fb.data_ptr(), None, self.valid.data_ptr())
        if rc != 0:
            raise RuntimeError("mbavo_eval_batch failed: %d" % rc)


def algorithmic_flops(probs, valid_pixels=None):
    """flops_alg(H,g) = PS*(363 + 48k) + PX*(2E + 12k + 13), counted from the
    reference source as written (every add/sub/mul/div/sqrt = 1)."""
    total = 0.0
    for i, p in enumerate(probs):
        px = p.F * p.K * p.P if valid_pixels is None else valid_pixels[i]
        E = synth.packed_len(p.k)
        total += px * p.S * (363 + 48 * p.k) + px * (2 * E + 12 * p.k + 13)
    return total


def algorithmic_bytes(probs, shard=None):
    """compulsory HBM bytes: images once (ref u8 + gradient 2 x f32 + current u8 per frame),
    keypoints (xy, z), pose tables, packed output blocks.  shard = (mode, rank, world): the bytes of that rank's share
    of the workload -- 'frames': keyframe images and keypoints in full, its own frames' current images; 'keypoints': a
    1/world band of every image and of the keypoints."""
    total = 0.0
    seen = set()
    mode, rank, world = shard if shard is not None else ("none", 0, 1)
    for p in probs:
        E = synth.packed_len(p.k)
        f0, f1 = ((p.F * rank) // world, (p.F * (rank + 1)) // world) if mode == "frames" else (0, p.F)
        band = 1.0 / world if mode == "keypoints" else 1.0
        for a in [p.ref, p.grad] + list(p.cur[f0:f1]):
            key = a.__array_interface__["data"][0]
            if key not in seen:
                seen.add(key)
                total += a.nbytes * band
        total += p.K * band * 24 + (f1 - f0) * p.S * (7 + 21 * p.k) * 8 + (f1 - f0) * E * 8
    return total
