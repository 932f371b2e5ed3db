"""Deterministic synthetic inputs for the blur-aware tracking path (numpy only).

Mirrors the inputs of the reference's module harness
(test/test_blur_aware_tracker_modules.cpp:24-81, 662-679) and the input
producers the tracker runs before the hot path (ImagePyramid.h:59-99,
Gradient.h:16-75).  Used by tests/ and bench.py to build problems; contains no
oracle or reference code.
"""
import numpy as np

PATTERN8 = np.array([-2, -2, 2, -2, -1, -1, 1, -1, 0, 0, 0, 1, -2, 2, 2, 2], dtype=np.int32)


def rpy_quat(roll, pitch, yaw):
    """xyzw quaternion of Transformation::setRollPitchYaw (Transformation.cpp:146-162)."""
    cr, sr = np.cos(0.5 * roll), np.sin(0.5 * roll)
    cp, sp = np.cos(0.5 * pitch), np.sin(0.5 * pitch)
    cy, sy = np.cos(0.5 * yaw), np.sin(0.5 * yaw)
    q = np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                  cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])
    return q / np.sqrt((q * q).sum())


_HARNESS_RPY = [(0.01, 0.01, 0.002), (0.02, 0.015, 0.0015), (0.03, 0.02, 0.001), (0.04, 0.025, 0.0005),
                (0.05, 0.03, 0.0), (0.05, 0.035, -0.0005), (0.07, 0.04, -0.001)]


def harness_spline(trans_scale=1.0, rot_scale=1.0, n_knots=7):
    """The 7-knot spline of create_spline (test/...modules.cpp:24-67): returns
    (knots_t [n,3], knots_R [n,4] xyzw).  Scales < 1 give the small motions used
    for tracking scenes."""
    kt = np.array([[5.0 * i, 5.0 * i, 0.0] for i in range(n_knots)]) * trans_scale
    kR = np.stack([rpy_quat(*(np.array(_HARNESS_RPY[i % 7]) * np.pi * rot_scale)) for i in range(n_knots)])
    return np.ascontiguousarray(kt), np.ascontiguousarray(kR)


def loop_spline(n_knots, radius=0.9, knots_per_turn=9.0, z_amp=0.25, rot_amp=0.012):
    """A bounded ground-truth trajectory for long sequences: the knots walk a circle of `radius` in the image plane (one turn
    every `knots_per_turn` knots) with a slow depth and roll / pitch / yaw oscillation at incommensurate rates, so that any
    number of frames stays in front of the same textured plane (the harness spline runs off it after ~15 frames at 640 x 480).
    Returns (knots_t [n,3], knots_R [n,4] xyzw); about the harness spline's image-plane speed at its 0.15 / 0.02 scales."""
    a = 2.0 * np.pi * np.arange(n_knots) / knots_per_turn
    kt = np.stack([radius * (1.0 - np.cos(a)), radius * np.sin(a), z_amp * np.sin(0.37 * a)], 1)
    kR = np.stack([rpy_quat(rot_amp * np.sin(0.61 * x), rot_amp * np.sin(0.43 * x + 1.0), 0.5 * rot_amp * np.sin(0.29 * x + 2.0))
                   for x in a])
    return np.ascontiguousarray(kt), np.ascontiguousarray(kR)


def quat_mul(a, b):
    """Hamilton product of xyzw quaternions (rotation a after b)."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def zigzag_spline(n_knots, trans_scale=0.1, rot_scale=0.02, period=8):
    """The module harness's trajectory family (create_spline, test/...modules.cpp:24-67: knots on the image-plane diagonal, its
    roll / pitch / yaw table) made bounded for long sequences: the diagonal walk 5 i is folded into a triangle wave of `period`
    knots, so the camera sweeps back and forth over at most 2.5 * period * trans_scale metres -- constant-velocity legs with a
    reversal every period / 2 knots, unlike loop_spline's circle."""
    i = np.arange(n_knots)
    tri = np.abs(((i + period // 2) % period) - period // 2).astype(np.float64)
    kt = np.stack([5.0 * tri, 5.0 * tri, np.zeros(n_knots)], 1) * trans_scale
    kR = np.stack([rpy_quat(*(np.array(_HARNESS_RPY[j % 7]) * np.pi * rot_scale)) for j in i])
    return np.ascontiguousarray(kt), np.ascontiguousarray(kR)


def tilted(kt, kR, pitch_deg=12.0, roll_deg=5.0):
    """The same camera path looking at the plane OBLIQUELY: every knot's rotation composed with a fixed tilt (camera frame), so the
    plane z = D of the plane frame is a tilted plane for the camera -- the depth varies across the image (frontend / sequence
    plane_depth_map follow the pose) and so does the flow a given motion causes."""
    tilt = rpy_quat(np.deg2rad(roll_deg), np.deg2rad(pitch_deg), 0.0)
    return kt, np.ascontiguousarray(np.stack([quat_mul(q, tilt) for q in kR]))


def trajectory(name, n_knots, trans_scale=0.15, rot_scale=0.02):
    """Ground-truth spline knots by name: "harness" (7 knots, runs off the texture after ~15 frames at 640 x 480), "loop",
    "zigzag" (the harness family, bounded), "loop_tilted" (the loop seen under a 12 degree pitch / 5 degree roll tilt)."""
    if name == "loop":
        return loop_spline(n_knots)
    if name == "zigzag":
        return zigzag_spline(n_knots)
    if name == "loop_tilted":
        return tilted(*loop_spline(n_knots))
    if name == "harness":
        return harness_spline(trans_scale, rot_scale, n_knots)
    raise ValueError(name)


def ramp_image(H=480, W=640):
    """create_uniform_image (test/...modules.cpp:69-81)."""
    return ((np.arange(W)[None, :] + np.arange(H)[:, None]) % 255).astype(np.uint8)


def noise_image(H=480, W=640, seed=1, smooth=3):
    """Band-limited uint8 texture with gradients everywhere."""
    rng = np.random.default_rng(seed)
    a = rng.random((H + 2 * smooth * 4, W + 2 * smooth * 4))
    for _ in range(smooth):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1) +
             np.roll(np.roll(a, 1, 0), 1, 1) + np.roll(np.roll(a, -1, 0), -1, 1) +
             np.roll(np.roll(a, 1, 0), -1, 1) + np.roll(np.roll(a, -1, 0), 1, 1)) / 9.0
    a = a[smooth * 4:smooth * 4 + H, smooth * 4:smooth * 4 + W]
    a = (a - a.min()) / (a.max() - a.min())
    return np.ascontiguousarray((a * 255.0).astype(np.uint8))


def texture_image(H=480, W=640, seed=1, octaves=(64, 32, 16, 8, 4)):
    """Multi-scale value noise (one bilinear-upsampled random lattice per octave, amplitude ~ cell
    size): has image gradients at every pyramid level, unlike white-ish noise."""
    rng = np.random.default_rng(seed)
    acc = np.zeros((H, W))
    yy, xx = np.mgrid[0:H, 0:W]
    for cell in octaves:
        gh, gw = H // cell + 2, W // cell + 2
        lat = rng.random((gh, gw))
        fy, fx = yy / cell, xx / cell
        y0, x0 = fy.astype(int), fx.astype(int)
        ty, tx = fy - y0, fx - x0
        ty, tx = ty * ty * (3 - 2 * ty), tx * tx * (3 - 2 * tx)
        v = (lat[y0, x0] * (1 - ty) * (1 - tx) + lat[y0, x0 + 1] * (1 - ty) * tx +
             lat[y0 + 1, x0] * ty * (1 - tx) + lat[y0 + 1, x0 + 1] * ty * tx)
        acc += v * cell
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    return np.ascontiguousarray((acc * 255.0).astype(np.uint8))


def shapes_image(H=480, W=640, blur=2):
    """Rectangles / triangles scene of generate_synthetic_data.cpp:11-125 (fg 255 on
    bg 0), lightly box-blurred so the gradient is non-zero near edges."""
    im = np.zeros((H, W), np.float64)
    sx, sy = W / 640.0, H / 480.0
    for (x, y, w, h) in [(300, 50, 50, 100), (250, 200, 100, 50), (400, 300, 100, 100),
                         (500, 50, 100, 100), (250, 300, 100, 100)]:
        im[int(y * sy):int((y + h) * sy) + 1, int(x * sx):int((x + w) * sx) + 1] = 255
    yy, xx = np.mgrid[0:H, 0:W]
    for tri in [((500, 50), (400, 150), (550, 250)), ((150, 300), (50, 450), (250, 400))]:
        (x0, y0), (x1, y1), (x2, y2) = [(px * sx, py * sy) for px, py in tri]
        d = (y1 - y2) * (x0 - x2) + (x2 - x1) * (y0 - y2)
        a = ((y1 - y2) * (xx - x2) + (x2 - x1) * (yy - y2)) / d
        b = ((y2 - y0) * (xx - x2) + (x0 - x2) * (yy - y2)) / d
        im[(a >= 0) & (b >= 0) & (a + b <= 1)] = 255
    for _ in range(blur):
        im = (im + np.roll(im, 1, 0) + np.roll(im, -1, 0) + np.roll(im, 1, 1) + np.roll(im, -1, 1)) / 5.0
    return np.ascontiguousarray(im.astype(np.uint8))


def image_gradients(img):
    """Central differences, interleaved [dx,dy] float32, zero 1-px border (Gradient.h:16-75)."""
    H, W = img.shape
    f = img.astype(np.float32)
    g = np.zeros((H, W, 2), np.float32)
    g[1:-1, 1:-1, 0] = 0.5 * (f[1:-1, 2:] - f[1:-1, :-2])
    g[1:-1, 1:-1, 1] = 0.5 * (f[2:, 1:-1] - f[:-2, 1:-1])
    return g


def pyramid(img, levels):
    """2x2 box with truncation (ImagePyramid.h:59-99); level l has size H0//2^l x W0//2^l."""
    out = [np.ascontiguousarray(img)]
    H0, W0 = img.shape
    for l in range(1, levels):
        p = out[-1].astype(np.uint16)
        Hl, Wl = H0 // (2 ** l), W0 // (2 ** l)
        s = p[0:2 * Hl:2, 0:2 * Wl:2] + p[0:2 * Hl:2, 1:2 * Wl:2] + p[1:2 * Hl:2, 0:2 * Wl:2] + p[1:2 * Hl:2, 1:2 * Wl:2]
        out.append(np.ascontiguousarray((s // 4).astype(np.uint8)))
    return out


def dense_keypoints(H, W, margin=0, z_lo=5.0, z_hi=10.0, seed=2, const_z=None):
    """Every pixel (inside `margin`) is a P=1 patch: xy [K,2] float64, z [K]."""
    ys, xs = np.mgrid[margin:H - margin, margin:W - margin]
    xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float64)
    K = xy.shape[0]
    if const_z is not None:
        z = np.full(K, float(const_z))
    else:
        z = np.random.default_rng(seed).uniform(z_lo, z_hi, K)
    return np.ascontiguousarray(xy), np.ascontiguousarray(z)


def semi_dense_keypoints(img, cell=30, thresh=25.0, margin=20, z_lo=5.0, z_hi=10.0, seed=2, const_z=None):
    """One keypoint per cell x cell grid cell: the max gradient-magnitude pixel if above
    `thresh` (the selection rule of FeatureDetectorSemiDense.cpp:16-59 /
    FeatureDetectorBase.cpp:49-91), kept `margin` px from the border."""
    H, W = img.shape
    g = image_gradients(img)
    mag = np.sqrt(g[..., 0] ** 2 + g[..., 1] ** 2)
    pts = []
    for y0 in range(0, H, cell):
        for x0 in range(0, W, cell):
            blk = mag[y0:y0 + cell, x0:x0 + cell]
            i = int(np.argmax(blk))
            by, bx = divmod(i, blk.shape[1])
            x, y = x0 + bx, y0 + by
            if blk[by, bx] > thresh and margin <= x < W - margin and margin <= y < H - margin:
                pts.append((float(x), float(y)))
    xy = np.array(pts, np.float64).reshape(-1, 2)
    K = xy.shape[0]
    z = np.full(K, float(const_z)) if const_z is not None else np.random.default_rng(seed).uniform(z_lo, z_hi, K)
    return np.ascontiguousarray(xy), np.ascontiguousarray(z)


def harness_keypoints(n=145, seed=7):
    """145 keypoints in [20,620)x[20,460), z in [20,45) (test/...modules.cpp:441-446),
    from a fixed seed instead of unseeded rand()."""
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(20, 620, n), rng.integers(20, 460, n)], 1).astype(np.float64)
    z = rng.uniform(20, 45, n)
    return np.ascontiguousarray(xy), np.ascontiguousarray(z)


def packed_len(k):
    nd = 6 * k + 1
    return nd * (nd + 1) // 2


def segment_start_index(t, t0, dt):
    """(int)((t - t0)/dt), truncation toward zero (SplineFunctor.h:13-19)."""
    return int((t - t0) / dt)


def pack_keyframe(img):
    """The packed keyframe of mbavo_pack_keyframe_u8 (mbavo_problem.grad_fp16 = 2) on the host: one uint32 per pixel, bits 0-7 the
    intensity, bits 8-16 / 23-31 the doubled central differences (Gradient.h:16-75: zero on the 1-pixel border) as 9-bit two's
    complement."""
    I = img.astype(np.int64)
    kx = np.zeros_like(I)
    ky = np.zeros_like(I)
    kx[1:-1, 1:-1] = I[1:-1, 2:] - I[1:-1, :-2]
    ky[1:-1, 1:-1] = I[2:, 1:-1] - I[:-2, 1:-1]
    w = I | ((kx & 0x1ff) << 8) | ((ky & 0x1ff) << 23)
    return np.ascontiguousarray(w.astype(np.uint32))
