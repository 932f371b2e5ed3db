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
        self.grad_fp16 = False  # True / 1: upload the gradient image as IEEE half pairs (BASELINE configs[4]); 2: the packed keyframe

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


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ v


def loop_spline(n_knots, dt=0.5, amp=(0.5, 0.35, 0.04), period=(2.1, 1.7, 3.3), rot_amp=(0.012, 0.01, 0.006)):
    """Control knots of a bounded ground-truth trajectory (a Lissajous loop in front of the textured plane, small periodic
    roll / pitch / yaw): any number of frames stays on the texture, unlike the harness spline's straight line."""
    kt = np.zeros((n_knots, 3))
    kR = np.zeros((n_knots, 4))
    for i in range(n_knots):
        t = i * dt
        kt[i] = [amp[a] * np.sin(2 * np.pi * t / period[a] + 0.7 * a) for a in range(3)]
        kR[i] = synth.rpy_quat(*[rot_amp[a] * np.sin(2 * np.pi * t / (period[a] * 1.3) + 1.1 * a) for a in range(3)])
    return np.ascontiguousarray(kt), np.ascontiguousarray(kR)


class _PairInfo:
    """What bench.py's accounting needs to know about one device-resident pair (no host copies of the images)."""

    def __init__(self, S, k, N, F, K, P, H, W, fmt=0):
        self.S, self.k, self.N, self.F, self.K, self.P, self.H, self.W = S, k, N, F, K, P, H, W
        # SURVEY.md 8(d), semi-dense: the compulsory bytes are the taps of the distinct tap locations, bounded above by the
        # gather figure 36 B per pixel-sample (2 x 2 u8 + 2 x 2 x 8 B gradient taps) + the current pixel -- and by the
        # whole images (keyframe u8 + gradient 2 x f32 + current u8), all of them this pair's own
        # (keyframe formats 1 / 2: half pairs 1 + 4 bytes per pixel, 20 per tap; packed words 4 and 16)
        img, tap = {0: (9, 36), 1: (5, 20), 2: (4, 16)}[int(fmt)]
        self.fmt = int(fmt)
        self.image_bytes_upper = min(H * W * img + F * H * W, F * K * P * (S * tap + 1))  # the gather bound: no reuse at all
        # until distinct_tap_bytes() has counted the pair's distinct tap locations the upper bound stands in
        self.image_bytes = self.image_bytes_upper
        self.distinct = None  # (distinct keyframe pixels tapped, distinct current pixels read, 128-byte lines touched)

    @property
    def pixel_samples(self):
        return self.F * self.K * self.P * self.S


class RenderedPairBatch:
    """BASELINE configs[2]/[3] as the configs describe them: B independent keyframe pairs = B consecutive frames of ONE
    synthetic blurred sequence (a textured plane, a camera on a ground-truth spline; generate_synthetic_data.cpp:127-214),
    every pair with its OWN keyframe image (the sharp rendering at the keyframe's time), its own gradient image, its own
    grid-selected keypoints with depths read from its own z-depth map, its own current image (the motion-blurred
    rendering one frame later, mbavo_synthesize_blur) and its own control knots (the ground-truth spline expressed in the
    keyframe's camera by left-multiplication -- the cumulative B-spline is left-invariant -- plus a small perturbation).
    Everything is rendered and detected on the GPU and stays there; `host_problem(b)` downloads one pair for the parity
    tests.  Same interface as DeviceWorkload (array, step, frame_blocks, valid, probs)."""

    def __init__(self, ctx, B, H=480, W=640, S=8, k=4, device="cuda:0", seed=1, huber=10.0, D=7.5, frame_dt=0.1, exp=0.04,
                 cell=30, thresh=4.0, perturb=2e-3, pairs=None, grad_fp16=False):
        import torch
        L = ctx.lib
        rng = np.random.default_rng(seed)
        self.B, self.k, self.S, self.H, self.W, self.device = B, k, S, H, W, device
        self.E = synth.packed_len(k)
        dtk, t0w, t_first = 0.5, 0.0, 0.55  # frame times t_first + i * frame_dt never straddle a knot (multiples of 0.5 +- exp)
        n_world = int((t_first + (B + 2) * frame_dt + exp) / dtk) + 5
        self.kt_w, self.kR_w = loop_spline(n_world, dtk)
        ktw, kRw = np.ascontiguousarray(self.kt_w.ravel()), np.ascontiguousarray(self.kR_w.ravel())
        intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0])
        self.intr, self.D = intr, D
        base = torch.from_numpy(synth.texture_image(H, W, seed=seed, octaves=(32, 16, 8, 4))).to(device)
        xs = torch.arange(W, dtype=torch.float64, device=device)[None, :].expand(H, W)
        ys = torch.arange(H, dtype=torch.float64, device=device)[:, None].expand(H, W)
        pat = torch.from_numpy(synth.PATTERN8).to(device)
        margin = 20
        self.keep = [base, pat]
        self.array = (capi.Problem * B)()
        self.probs, self._host = [], []
        own = range(B) if pairs is None else pairs  # (a rank may build only its own pairs; the others stay zero-filled)
        own = set(own)
        cap_kp = (H // cell + 1) * (W // cell + 1)
        # pair b's knot perturbation is the b-th draw whoever renders it (a rank that builds only its own pairs, or a checker
        # that re-renders single pairs, sees the pairs the whole batch holds)
        perts = [rng.normal(0, perturb, (4, 3)) for _ in range(B)]
        for b in range(B):
            if b not in own:
                self.probs.append(_PairInfo(S, k, 4, 1, 0, 8, H, W, grad_fp16))
                self._host.append(None)
                continue
            tk, tc = t_first + b * frame_dt, t_first + (b + 1) * frame_dt
            pk, qk = np.zeros(3), np.zeros(4)
            capi.check(L.mbavo_spline_get_pose(4, t0w, dtk, capi.dp(ktw), capi.dp(kRw), n_world, float(tk), capi.dp(pk), capi.dp(qk),
                                               None, None), "mbavo_spline_get_pose")
            ref = torch.empty(H * W, dtype=torch.uint8, device=device)
            cur = torch.empty(H * W, dtype=torch.uint8, device=device)
            for (t, e, ns, dst) in ((tk, 0.0, 2, ref), (tc, exp, 8, cur)):
                capi.check(L.mbavo_synthesize_blur(base.data_ptr(), H, W, float(D), capi.dp(intr), 4, t0w, dtk, capi.dp(ktw),
                                                   capi.dp(kRw), n_world, float(t), float(e), ns, dst.data_ptr(), None),
                           "mbavo_synthesize_blur")
            if int(grad_fp16) == 2:  # packed keyframe: intensity + both differences in one word per pixel
                grad = torch.empty(H * W, dtype=torch.int32, device=device)
                capi.check(L.mbavo_pack_keyframe_u8(ref.data_ptr(), H, W, grad.data_ptr(), None), "mbavo_pack_keyframe_u8")
            elif grad_fp16:  # IEEE half pairs (lossless for central differences of an 8-bit image)
                grad = torch.empty(H * W * 2, dtype=torch.float16, device=device)
                capi.check(L.mbavo_image_gradients_u8_half(ref.data_ptr(), H, W, grad.data_ptr(), None), "mbavo_image_gradients_u8_half")
            else:
                grad = torch.empty(H * W * 2, dtype=torch.float32, device=device)
                capi.check(L.mbavo_image_gradients_u8(ref.data_ptr(), H, W, grad.data_ptr(), None), "mbavo_image_gradients_u8")
            # z-depth of the plane z = D (plane frame) in the keyframe camera (camera -> plane pose (qk, pk))
            x, y, z, w = qk
            r2 = (2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y))
            rz = (xs - intr[2]) / intr[0] * r2[0] + (ys - intr[3]) / intr[1] * r2[1] + r2[2]
            depth = ((D - pk[2]) / rz).to(torch.float32).contiguous()
            torch.cuda.synchronize()
            xy = torch.empty(cap_kp * 2, dtype=torch.float64, device=device)
            kz = torch.empty(cap_kp, dtype=torch.float64, device=device)
            cnt = C.c_int(0)
            capi.check(L.mbavo_detect_semidense(ctx.handle, ref.data_ptr(), H, W, 0, H, W, cell, cell, float(thresh),
                                                depth.data_ptr(), xy.data_ptr(), kz.data_ptr(), cap_kp, C.byref(cnt)),
                       "mbavo_detect_semidense")
            K = cnt.value
            # keep the keypoints whose patch stays `margin` pixels inside the image (as the semi-dense builder does)
            xyv = xy[:2 * K].view(K, 2)
            ok = (xyv[:, 0] >= margin) & (xyv[:, 0] < W - margin) & (xyv[:, 1] >= margin) & (xyv[:, 1] < H - margin)
            xy = xyv[ok].contiguous().view(-1)
            kz = kz[:K][ok].contiguous()
            K = int(kz.shape[0])
            # knots of the segment that holds the current frame's exposure, in the keyframe's camera
            idx = int((tc - exp * 0.5 - t0w) / dtk)
            assert int((tc + exp * 0.5 - t0w) / dtk) == idx and idx + 4 <= n_world
            qk_inv = np.array([-qk[0], -qk[1], -qk[2], qk[3]])
            kt = np.stack([_qrot(qk_inv, self.kt_w[idx + i] - pk) for i in range(4)])
            kR = np.stack([_qmul(qk_inv, self.kR_w[idx + i]) for i in range(4)])
            kR /= np.linalg.norm(kR, axis=1, keepdims=True)
            kt_gt = kt.copy()
            kt = kt + perts[b]
            t0 = t0w + idx * dtk
            capt, expt = torch.tensor([tc], dtype=torch.float64, device=device), torch.tensor([exp], dtype=torch.float64, device=device)
            dkt, dkR = torch.from_numpy(kt.ravel().copy()).to(device), torch.from_numpy(kR.ravel().copy()).to(device)
            cur_ptrs = torch.tensor([cur.data_ptr()], dtype=torch.int64, device=device)
            start = np.array([synth.segment_start_index(tc, t0, dtk)], np.int32)
            assert start[0] == 0
            self.keep += [ref, cur, grad, xy, kz, capt, expt, dkt, dkR, cur_ptrs, start]
            q = self.array[b]
            q.S, q.F, q.K, q.P, q.N, q.H, q.W = S, 1, K, 8, 4, H, W
            q.d_ref_img, q.d_ref_dIxy, q.d_cur_imgs = ref.data_ptr(), grad.data_ptr(), cur_ptrs.data_ptr()
            q.d_kp_xy, q.kp_stride, q.d_kp_z, q.d_pattern = xy.data_ptr(), 2, kz.data_ptr(), pat.data_ptr()
            q.d_outlier, q.num_bad = None, 0
            for i in range(4):
                q.intrinsics[i] = float(intr[i])
            q.d_cap_time, q.d_exp_time, q.t0, q.dt = capt.data_ptr(), expt.data_ptr(), t0, dtk
            q.d_knots_t, q.d_knots_R = dkt.data_ptr(), dkR.data_ptr()
            q.h_start_idx = start.ctypes.data_as(C.POINTER(C.c_int))
            q.huber_a, q.grad_fp16 = huber, int(grad_fp16)
            self.probs.append(_PairInfo(S, k, 4, 1, K, 8, H, W, grad_fp16))
            self._host.append(dict(ref=ref, cur=cur, grad=grad, xy=xy, kz=kz, kt=kt, kR=kR, kt_gt=kt_gt, t0=t0, cap=tc, exp=exp,
                                   huber=huber, pk=pk, qk=qk, dkt=dkt, dkR=dkR))
        self.nbf = B
        self.frame_blocks = torch.zeros(self.nbf * self.E, dtype=torch.float64, device=device)
        self.valid = torch.zeros(self.nbf, dtype=torch.float64, device=device)
        torch.cuda.synchronize()

    def count_distinct_taps(self, ctx, pairs=None):
        """the COMPULSORY bytes of a semi-dense pair are the bytes of its DISTINCT tap locations (36 B per
        pixel-sample is the no-reuse upper bound).  Counts them on the host for the pairs' actual keypoints and knots: patch
        centre at blur sample S/2 (compute_local_patches_xy.cu:19-49), the truncated pixel (A3), for every blur sample the warp
        through its pose (compute_pixel_intensity.h:117-144) and the 2 x 2 tap window anchored at min(floor, size - 2) -- plain
        numpy (the figure is a count of locations; a last-place difference in a coordinate moves no window).  Sets, per pair,
        probs[b].distinct = (keyframe pixels, current pixels, 128-byte lines of the row-major images) and
        probs[b].image_bytes = the distinct-tap bytes in the pair's keyframe format."""
        L = ctx.lib
        H, W, S = self.H, self.W, self.S
        fx, fy, cx, cy = [float(v) for v in self.intr]
        pat = synth.PATTERN8.reshape(-1, 2).astype(np.int64)
        per_px = {0: (1, 8), 1: (1, 4), 2: (0, 4)}  # keyframe bytes per distinct pixel: (u8 image, gradient / packed image)
        for b in (range(self.B) if pairs is None else pairs):
            h = self._host[b]
            if h is None:
                continue
            xy = h["xy"].cpu().numpy().reshape(-1, 2)
            kz = h["kz"].cpu().numpy()
            kt, kR = np.ascontiguousarray(h["kt"].ravel()), np.ascontiguousarray(h["kR"].ravel())
            poses = []
            for i in range(S):
                t = h["cap"] - h["exp"] * 0.5 + i * h["exp"] / (S - 1 + 1e-8)
                pp, qq = np.zeros(3), np.zeros(4)
                capi.check(L.mbavo_spline_get_pose(self.k, h["t0"], 0.5, capi.dp(kt), capi.dp(kR), 4, float(t), capi.dp(pp), capi.dp(qq),
                                                   None, None), "mbavo_spline_get_pose")
                x, y, z, w = qq
                R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                poses.append((R, pp))
            Rm, tm = poses[S // 2]
            P3r = np.stack([kz * (xy[:, 0] - cx) / fx, kz * (xy[:, 1] - cy) / fy, kz], 1)
            P3c = (P3r - tm) @ Rm                       # R^T (P - t)
            cen = np.stack([P3c[:, 0] / P3c[:, 2] * fx + cx, P3c[:, 1] / P3c[:, 2] * fy + cy], 1)
            px = (cen[:, None, 0] + pat[None, :, 0]).astype(np.int64)   # truncation toward zero, as (int)
            py = (cen[:, None, 1] + pat[None, :, 1]).astype(np.int64)
            inb = (px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)
            d = np.broadcast_to(kz[:, None], px.shape)[inb]
            px, py = px[inb], py[inb]
            cur_ids = np.unique(py * W + px)
            rx, ry = (px - cx) / fx, (py - cy) / fy
            zh = 1.0 / np.sqrt(1.0 + rx * rx + ry * ry)
            ray = np.stack([rx * zh, ry * zh, zh], 1)
            ids = []
            for R, t in poses:
                rr = ray @ R.T
                lam = (d - t[2]) / rr[:, 2]
                Pr = rr * lam[:, None] + t
                u, v = Pr[:, 0] / Pr[:, 2] * fx + cx, Pr[:, 1] / Pr[:, 2] * fy + cy
                ok = (u >= 0) & (u <= W - 1) & (v >= 0) & (v <= H - 1)
                x0 = np.minimum(np.floor(u[ok]).astype(np.int64), W - 2)
                y0 = np.minimum(np.floor(v[ok]).astype(np.int64), H - 2)
                for dy in (0, 1):
                    for dx in (0, 1):
                        ids.append((y0 + dy) * W + (x0 + dx))
            ref_ids = np.unique(np.concatenate(ids)) if ids else np.zeros(0, np.int64)
            fmt = self.probs[b].fmt
            b_img, b_grad = per_px[fmt]
            lines = len(np.unique(cur_ids // 128)) + (len(np.unique(ref_ids // 128)) if b_img else 0) + len(np.unique(ref_ids * b_grad // 128))
            self.probs[b].distinct = (int(len(ref_ids)), int(len(cur_ids)), int(lines))
            self.probs[b].image_bytes = int(len(ref_ids) * (b_img + b_grad) + len(cur_ids))
        return self

    def reset_knots(self):
        """Initial control knots back into the device buffers (mbavo_lm_batch updates them in place)."""
        import torch
        for h in self._host:
            if h is not None:
                h["dkt"].copy_(torch.from_numpy(h["kt"].ravel().copy()))
                h["dkR"].copy_(torch.from_numpy(h["kR"].ravel().copy()))
        torch.cuda.synchronize()

    def host_problem(self, b):
        """Pair b as a numpy Prob (images downloaded): what the oracle is given in the parity tests."""
        h = self._host[b]
        H, W = self.H, self.W
        return Prob(h["ref"].cpu().numpy().reshape(H, W), [h["cur"].cpu().numpy().reshape(H, W)],
                    h["xy"].cpu().numpy().reshape(-1, 2), h["kz"].cpu().numpy(), synth.PATTERN8, self.intr, self.S, self.k, 4,
                    [h["cap"]], [h["exp"]], h["t0"], 0.5, h["kt"], h["kR"], h["huber"],
                    grad=h["grad"].float().cpu().numpy().reshape(H, W, 2))

    def step(self, ctx, with_hessian=True, out=None, merged=True):
        _step(self, ctx, with_hessian, out, merged)


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
            if int(p.grad_fp16) == 2:
                grad = up(synth.pack_keyframe(p.ref))
            elif p.grad_fp16:
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
            q.grad_fp16 = int(p.grad_fp16)
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

    def step(self, ctx, with_hessian=True, out=None, merged=True):
        """One GN-iteration evaluation of every problem (asynchronous on the context's stream); `out` replaces the
        default output tensor (double buffering under an asynchronous all-reduce)."""
        _step(self, ctx, with_hessian, out, merged)


def _step(w, ctx, with_hessian, out, merged):
    """One pass of the hot path over the workload's problem list.  With H / g the pass ends in the reference's unit
    (evaluate_cost_hessian_gradient, spline_update_step.cpp:97-241): every problem's merged system [cost | g (6N) | H (6N x 6N)]
    in `w.systems` on the device (mbavo_eval_batch_merged: the merge of :232-239 is part of the finalize step where every
    problem has one frame and N == k, a kernel behind it otherwise) next to the packed frame blocks; merged=False keeps
    mbavo_eval_batch alone (packed blocks only: what the batched LM and the multi-GPU collectives consume)."""
    import torch
    fb = w.frame_blocks if out is None else out
    if with_hessian and merged:
        if getattr(w, "systems", None) is None:
            w.systems = torch.zeros(sum(1 + 6 * int(w.array[b].N) + 36 * int(w.array[b].N) ** 2 for b in range(w.B)), dtype=torch.float64,
                                    device=fb.device)
        rc = ctx.lib.mbavo_eval_batch_merged(ctx.handle, w.B, w.array, w.k, fb.data_ptr(), w.systems.data_ptr(), None, w.valid.data_ptr())
    else:
        rc = ctx.lib.mbavo_eval_batch(ctx.handle, w.B, w.array, w.k, 1 if with_hessian else 0, fb.data_ptr(), None, w.valid.data_ptr())
    if rc != 0:
        raise RuntimeError("mbavo_eval_batch%s failed: %d" % ("_merged" if with_hessian and merged else "", rc))


def algorithmic_flops(probs, valid_pixels=None):
    """flops_alg(H,g) = PS*(363 + 48k) + PX*(2E + 12k + 13), counted from the
    reference source as written (every add/sub/mul/div/sqrt = 1)."""
    total = 0.0
    for i, p in enumerate(probs):
        px = p.F * p.K * p.P if valid_pixels is None else valid_pixels[i]
        E = synth.packed_len(p.k)
        total += px * p.S * (363 + 48 * p.k) + px * (2 * E + 12 * p.k + 13)
    return total


def algorithmic_bytes(probs, shard=None, upper=False):
    """compulsory HBM bytes: images once (ref u8 + gradient 2 x f32 + current u8 per frame),
    keypoints (xy, z), pose tables, packed output blocks.  shard = (mode, rank, world): the bytes of that rank's share
    of the workload -- 'frames': keyframe images and keypoints in full, its own frames' current images; 'keypoints': a
    1/world band of every image and of the keypoints; 'pairs': the pairs b % world == rank, whole."""
    total = 0.0
    seen = set()
    mode, rank, world = shard if shard is not None else ("none", 0, 1)
    for b, p in enumerate(probs):
        if mode == "pairs" and b % world != rank:
            continue
        E = synth.packed_len(p.k)
        f0, f1 = ((p.F * rank) // world, (p.F * (rank + 1)) // world) if mode == "frames" else (0, p.F)
        band = 1.0 / world if mode == "keypoints" else 1.0
        if hasattr(p, "image_bytes"):  # a device-resident pair with its own images (RenderedPairBatch)
            # (its distinct tap locations once count_distinct_taps() has run -- SURVEY 8(d)'s compulsory figure; `upper`: the
            # gather bound, no reuse at all)
            total += (p.image_bytes_upper if upper else p.image_bytes) * band
        for a in ([] if hasattr(p, "image_bytes") else [p.ref, p.grad] + list(p.cur[f0:f1])):
            key = a.__array_interface__["data"][0]
            if key not in seen:
                seen.add(key)
                total += a.nbytes * band
        total += p.K * band * 24 + (f1 - f0) * p.S * (7 + 21 * p.k) * 8 + (f1 - f0) * E * 8
    return total
