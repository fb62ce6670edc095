"""Synthetic photoconsistent frame pairs (numpy only; no GPU, no reference, no datasets).

Neither KITTI nor ScanNet is available on the build or the GPU box, so
tests and bench.py use a procedurally textured, piecewise-planar "corridor" scene whose
depth is analytic in BOTH views:

    ground  y = h,   far wall  z = zf,   side walls  x = -wl, x = +wr      (target frame)

Camera convention is the reference's (KITTI cam: x right, y down, z forward).  The pose
6-vector ``[tx,ty,tz,rx,ry,rz]`` follows the reference's hot path: the warp is called with
``-pose`` (train_mono.py:69, helpers.py:11), i.e. a target-frame point X_t is seen by the
source camera at ``X_s = R(-r) X_t + (-t)`` with ``R = Rx Ry Rz`` (models/stn.py:81-116).

Both images are rendered by evaluating one smooth texture ``tex(X_t)`` at the analytic
ray/plane intersection, so the pair is exactly photoconsistent at the ground-truth pose up
to bilinear-interpolation error and the added sensor noise.

Units are the reference's network units (metres / 30, run_sequential_optimization.py:224).
"""
from __future__ import annotations

import numpy as np

# KITTI cam2 intrinsics scaled to 640x192 (create_kitti_odometry_data.py:62-63,92-93)
KITTI_K = np.array([[369.2, 0.0, 314.3], [0.0, 367.0, 95.0], [0.0, 0.0, 1.0]], dtype=np.float64)
KITTI_MIN_DEPTH, KITTI_MAX_DEPTH = 0.06, 2.67  # run_mono_exps_kitti.sh:5


def scaled_K(H: int, W: int) -> np.ndarray:
    """KITTI-like intrinsics rescaled from 640x192 to (W,H)."""
    K = KITTI_K.copy()
    K[0] *= W / 640.0
    K[1] *= H / 192.0
    return K


def euler_R(r) -> np.ndarray:
    """R = Rx(rx) Ry(ry) Rz(rz)  (models/stn.py:81-116)."""
    x, y, z = [float(v) for v in r]
    cx, sx, cy, sy, cz, sz = np.cos(x), np.sin(x), np.cos(y), np.sin(y), np.cos(z), np.sin(z)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def pose_to_T(pose) -> np.ndarray:
    """3x4 target->source transform used by the warp: pose_vec2mat(-pose) (stn.py:143-158)."""
    pose = np.asarray(pose, dtype=np.float64)
    return np.concatenate([euler_R(-pose[3:6]), (-pose[0:3]).reshape(3, 1)], axis=1)


def R_to_euler(R) -> np.ndarray:
    """Inverse of euler_R for |ry| < pi/2:  R = Rx(a) Ry(b) Rz(c)."""
    b = np.arcsin(np.clip(R[0, 2], -1.0, 1.0))
    a = np.arctan2(-R[1, 2], R[2, 2])
    c = np.arctan2(-R[0, 1], R[0, 0])
    return np.array([a, b, c])


def T_to_pose(T) -> np.ndarray:
    """Inverse of pose_to_T."""
    T = np.asarray(T, dtype=np.float64)
    return np.concatenate([-T[:, 3], -R_to_euler(T[:, :3])])


def invert_pose(pose) -> np.ndarray:
    """Pose 6-vector of the reversed pair (exact SE(3) inverse; the reference approximates it by
    the negated vector, optimizer.py:168)."""
    T = pose_to_T(pose)
    R, t = T[:, :3], T[:, 3]
    return T_to_pose(np.concatenate([R.T, (-R.T @ t).reshape(3, 1)], axis=1))


class _Texture:
    """Smooth multi-octave sinusoid texture over R^3 -> [0,1]^3, seeded."""

    def __init__(self, seed: int, octaves=(0.45, 0.22, 0.11, 0.055, 0.03)):
        rng = np.random.default_rng(seed)
        self.waves = []
        for lam in octaves:
            for _ in range(3):
                d = rng.normal(size=3)
                d /= np.linalg.norm(d)
                f = d / lam
                ph = rng.uniform(0, 2 * np.pi, size=3)
                amp = rng.uniform(0.5, 1.0, size=3) * (lam / octaves[0]) ** 0.6
                self.waves.append((f, ph, amp))
        self.norm = sum(w[2] for w in self.waves)

    def __call__(self, X: np.ndarray) -> np.ndarray:
        """X [...,3] -> rgb [3,...]"""
        out = np.zeros((3,) + X.shape[:-1])
        for f, ph, amp in self.waves:
            s = 2 * np.pi * (X @ f)
            for c in range(3):
                out[c] += amp[c] * np.sin(s + ph[c])
        out = 0.5 + 0.45 * out / self.norm[:, None, None] * 2.2
        return np.clip(out, 0.0, 1.0)


def _rays(H, W, K):
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    pix = np.stack([u, v, np.ones_like(u)], -1)
    return pix @ np.linalg.inv(K).T  # [H,W,3]


def _room_depth(rays, R, t, room):
    """Depth along ``rays`` (camera frame X_c = R X_t + t) to the convex room given in the target frame."""
    h, zf, wl, wr = room
    planes = [((0, 1, 0), h), ((0, 0, 1), zf), ((-1, 0, 0), wl), ((1, 0, 0), wr)]
    best = np.full(rays.shape[:2], np.inf)
    for n, d in planes:
        n = R @ np.asarray(n, dtype=np.float64)
        dd = d + n @ t
        den = rays @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            z = np.where(den > 1e-9, dd / den, np.inf)
        z = np.where(z > 0, z, np.inf)
        best = np.minimum(best, z)
    return best


def _rays_sampler(H, W, K):
    """rays through the positions at which the reference's warp reads SOURCE pixel (x, y): it samples the source image at
    ix = u W/(W-1) - 1/2 for the projected camera coordinate u (models/stn.py:198-231,266; grid_sample with align_corners=False
    on coordinates normalised by W-1), so source pixel x stands for the camera coordinate u = (x + 1/2)(W-1)/W"""
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    pix = np.stack([(u + 0.5) * (W - 1) / W, (v + 0.5) * (H - 1) / H, np.ones_like(u)], -1)
    return pix @ np.linalg.inv(K).T


def make_pair(H=192, W=640, seed=0, pose_gt=None, noise=0.003, K=None,
              room=(0.055, 1.5, 0.28, 0.33), dtype=np.float32, sampler_consistent=False):
    """One photoconsistent (target, source) pair.

    Returns dict: tgt,src [3,H,W]; depth_t, depth_s [H,W]; K [3,3]; pose_gt [6]
    (all ``dtype``).  Deterministic in (H,W,seed,pose_gt,noise,room).

    sampler_consistent: the SOURCE image and depth map are rendered through the reference's own sampling model (_rays_sampler),
    so that warping them with the true pose and depth reproduces the target up to bilinear interpolation and the minimiser of the
    reference's residual IS the scene's true pose (with the plain pinhole rendering it is offset by up to half a pixel of flow,
    SURVEY 8a row a5).  Such a source is only consistent in the sampled role: use it for directed pairs, not as the target of an
    inverse pair.
    """
    K = scaled_K(H, W) if K is None else np.asarray(K, dtype=np.float64)
    rng = np.random.default_rng(1000 + seed)
    if pose_gt is None:
        pose_gt = np.array([0.003, -0.002, 0.033, 0.002, -0.004, 0.0015]) * rng.uniform(0.7, 1.3, size=6)
    pose_gt = np.asarray(pose_gt, dtype=np.float64)
    T = pose_to_T(pose_gt)
    R, t = T[:, :3], T[:, 3]
    tex = _Texture(seed)
    rays = _rays(H, W, K)
    rays_s = _rays_sampler(H, W, K) if sampler_consistent else rays
    d_t = _room_depth(rays, np.eye(3), np.zeros(3), room)
    d_s = _room_depth(rays_s, R, t, room)
    X_t = rays * d_t[..., None]                      # target-frame points seen by target
    X_s = rays_s * d_s[..., None]                    # source-frame points seen by source
    X_s_in_t = (X_s - t) @ R                         # R^T (X_s - t)
    tgt = tex(X_t)
    src = tex(X_s_in_t)
    if noise > 0:
        tgt = np.clip(tgt + rng.normal(scale=noise, size=tgt.shape), 0, 1)
        src = np.clip(src + rng.normal(scale=noise, size=src.shape), 0, 1)
    return dict(tgt=tgt.astype(dtype), src=src.astype(dtype), depth_t=d_t.astype(dtype),
                depth_s=d_s.astype(dtype), K=K.astype(dtype), pose_gt=pose_gt.astype(dtype))


def perturb_pose(pose_gt, seed=0, sigma_t=0.001, sigma_r=0.0003):
    """Initial pose = GT + N(0, sigma_t^2) on translation, N(0, sigma_r^2) on rotation.

    Defaults model a PoseNet-quality initialisation (~3 % of the 1 m/frame motion, ~0.02 deg): the
    single-scale photometric basin is ~1-2 px of flow, and SURVEY §8d's looser 0.01/0.002 suggestion
    (7+ px of flow on the near ground plane) is outside it without a coarse-to-fine pyramid."""
    rng = np.random.default_rng(2000 + seed)
    p = np.asarray(pose_gt, dtype=np.float64).copy()
    p[:3] += rng.normal(scale=sigma_t, size=3)
    p[3:] += rng.normal(scale=sigma_r, size=3)
    return p.astype(np.asarray(pose_gt).dtype)


def depth_to_sigmoid_disp(depth, min_depth=KITTI_MIN_DEPTH, max_depth=KITTI_MAX_DEPTH):
    """Inverse of disp_to_depth (utils/learning_helpers.py:77-86, :89-98)."""
    min_disp, max_disp = 1.0 / max_depth, 1.0 / min_depth
    return (1.0 / depth - min_disp) / (max_disp - min_disp)


def make_batch(N, H=192, W=640, seed0=0, noise=0.003, dtype=np.float32, both_directions=False, sampler_consistent=False):
    """N directed pairs stacked: tgt,src [N,3,H,W]; depth_t,depth_s [N,1,H,W]; K [N,3,3];
    pose_gt, pose_init [N,6].  With ``both_directions`` pair 2i+1 is pair 2i reversed
    (the reference's fwd/inv stacking, train_mono.py:54-62) with pose_gt negated."""
    out = {k: [] for k in ("tgt", "src", "depth_t", "depth_s", "K", "pose_gt", "pose_init")}
    i = 0
    while len(out["tgt"]) < N:
        p = make_pair(H, W, seed=seed0 + i, noise=noise, dtype=dtype, sampler_consistent=sampler_consistent)
        init = perturb_pose(p["pose_gt"], seed=seed0 + i)
        for flip in ((False, True) if both_directions else (False,)):
            if len(out["tgt"]) >= N:
                break
            a, b, da, db = ("src", "tgt", "depth_s", "depth_t") if flip else ("tgt", "src", "depth_t", "depth_s")
            out["tgt"].append(p[a]); out["src"].append(p[b])
            out["depth_t"].append(p[da][None]); out["depth_s"].append(p[db][None])
            out["K"].append(p["K"])
            out["pose_gt"].append(invert_pose(p["pose_gt"]).astype(dtype) if flip else p["pose_gt"])
            out["pose_init"].append(invert_pose(init).astype(dtype) if flip else init)
        i += 1
    return {k: np.ascontiguousarray(np.stack(v)) for k, v in out.items()}


def make_sequence(T, H=192, W=640, seed=0, noise=0.003, dtype=np.float32, room=(0.055, 12.0, 0.28, 0.33)):
    """T frames of ONE textured corridor seen from a moving camera (frame t+1 is the source of frame t and the target of the next
    window, as in the reference's sequence loaders).  Returns dict: frames [T,3,H,W], depths [T,1,H,W], K [3,3],
    pose_gt [T-1,6] (frame t -> t+1, the convention of make_pair), init [T-1,2,6] = PoseNet-quality initial poses of the forward
    and inverse directed pair of every window."""
    K = scaled_K(H, W)
    rng = np.random.default_rng(3000 + seed)
    tex = _Texture(seed)
    rays = _rays(H, W, K)
    Rw, tw = np.eye(3), np.zeros(3)                       # world (= frame 0) -> camera t
    frames, depths, rel, init = [], [], [], []
    for t in range(T):
        d = _room_depth(rays, Rw, tw, room)
        Xw = (rays * d[..., None] - tw) @ Rw             # R^T (X_c - t)
        img = tex(Xw)
        if noise > 0:
            img = np.clip(img + rng.normal(scale=noise, size=img.shape), 0, 1)
        frames.append(img.astype(dtype)); depths.append(d[None].astype(dtype))
        if t < T - 1:
            p = np.array([0.002, -0.001, 0.033, 0.001, -0.003, 0.001]) + rng.normal(scale=[3e-4, 3e-4, 2e-3, 5e-4, 1e-3, 5e-4])
            rel.append(p)
            f0 = perturb_pose(p, seed * 1000 + t)
            init.append(np.stack([f0, invert_pose(f0)]))
            Tm = pose_to_T(p)
            Rw, tw = Tm[:, :3] @ Rw, Tm[:, :3] @ tw + Tm[:, 3]
    return dict(frames=np.stack(frames), depths=np.stack(depths), K=K.astype(dtype), pose_gt=np.stack(rel).astype(dtype),
                init=np.stack(init).astype(dtype))
