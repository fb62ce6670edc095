"""ctypes front-end of the CPU oracle (oracle/tcsfm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
# TCSFM_ORACLE_BUILD_DIR: load the libraries from another build directory (the sanitizer leg, scripts/run_asan.sh: _build_asan)
_BUILD = os.environ.get("TCSFM_ORACLE_BUILD_DIR") or os.path.join(_DIR, "_build")
MAXP = 7


class OrcOpts(C.Structure):
    _fields_ = [("nparam", C.c_int), ("automask", C.c_int), ("param", C.c_int), ("solver", C.c_int),
                ("n_iters", C.c_int), ("w_l1", C.c_double), ("w_ssim", C.c_double), ("w_dc", C.c_double),
                ("irls_eps", C.c_double), ("lambda0", C.c_double), ("lambda_up", C.c_double),
                ("lambda_down", C.c_double), ("lambda_min", C.c_double), ("prior_scale", C.c_double), ("w_pose_consist", C.c_double), ("w_smooth", C.c_double)]


class LinOut(C.Structure):
    _fields_ = [("H", C.c_double * (MAXP * MAXP)), ("g", C.c_double * MAXP), ("cost", C.c_double),
                ("cost_photo", C.c_double), ("cost_dc", C.c_double), ("n_mask", C.c_double)]


def default_opts(**kw) -> OrcOpts:
    o = OrcOpts(nparam=6, automask=1, param=0, solver=0, n_iters=4, w_l1=0.15, w_ssim=0.85, w_dc=0.0,
                irls_eps=1e-3, lambda0=1e-4, lambda_up=10.0, lambda_down=0.1, lambda_min=1e-5, prior_scale=1.0, w_pose_consist=0.0, w_smooth=0.0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def build(force: bool = False) -> None:
    """Compile both oracle libraries with gcc (seconds)."""
    if os.environ.get("TCSFM_ORACLE_BUILD_DIR"):
        return          # a pre-built alternative (sanitizer build): use it as it is
    if force or not all(os.path.exists(os.path.join(_BUILD, f"liboracle_{p}.so")) for p in ("f32", "f64")) or \
            os.path.getmtime(os.path.join(_DIR, "tcsfm_oracle.c")) > os.path.getmtime(os.path.join(_BUILD, "liboracle_f64.so")):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))


class Oracle:
    """precision 'f64' (the oracle proper) or 'f32' (diagnostic fp32 twin)."""

    def __init__(self, precision: str = "f64"):
        build()
        self.dt = {"f64": np.float64, "f32": np.float32}[precision]
        self.lib = C.CDLL(os.path.join(_BUILD, f"liboracle_{precision}.so"))
        assert self.lib.orc_sizeof_real() == np.dtype(self.dt).itemsize
        self.lib.orc_cost.restype = C.c_double
        self.lib.orc_scale_recovery.restype = C.c_double

    # -- helpers ---------------------------------------------------------
    def _r(self, a):
        return np.ascontiguousarray(a, dtype=self.dt)

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def _d(a):
        return np.ascontiguousarray(a, dtype=np.float64)

    @staticmethod
    def _forced(bits, decide):
        """the engine's decision trace (Engine.trace_*) as C arrays: bits uint16 (bit 0 mask, bit 1 warp validity, 2-3 bilinear cell parity, 4-11 sign codes), decide int32"""
        b = None if bits is None else np.ascontiguousarray(bits, dtype=np.uint16)
        d = None if decide is None else np.ascontiguousarray(decide, dtype=np.int32)
        return b, d

    # -- SE(3) / pose utilities (double) -----------------------------------
    def pose_to_T(self, pose):
        pose = self._d(pose); T = np.zeros(12)
        self.lib.orc_pose_to_T(self._p(pose), self._p(T))
        return T.reshape(3, 4)

    def T_to_pose(self, T):
        T = self._d(T).reshape(12); p = np.zeros(6)
        self.lib.orc_T_to_pose(self._p(T), self._p(p))
        return p

    def se3_exp(self, xi):
        xi = self._d(xi); T = np.zeros(12)
        self.lib.orc_se3_exp(self._p(xi), self._p(T))
        return T.reshape(3, 4)

    def se3_log(self, T):
        T = self._d(T).reshape(12); xi = np.zeros(6)
        self.lib.orc_se3_log(self._p(T), self._p(xi))
        return xi

    def euler_left_jacobian(self, pose):
        pose = self._d(pose); A = np.zeros(36)
        self.lib.orc_euler_left_jacobian(self._p(pose), self._p(A))
        return A.reshape(6, 6)

    # -- residual pieces ---------------------------------------------------
    def disp_to_depth(self, disp, min_depth, max_depth):
        d = self._r(disp); s = np.empty_like(d); z = np.empty_like(d)
        self.lib.orc_disp_to_depth(C.c_int(d.size), self._p(d), C.c_double(min_depth), C.c_double(max_depth),
                                   self._p(s), self._p(z))
        return s, z

    def warp(self, src, depth_t, depth_s, pose, K, log_scale=0.0, T=None):
        """inverse_warp2 for one pair: src [3,H,W], depth [H,W] -> rec [3,H,W], valid, proj_depth, comp_depth [H,W]."""
        src, depth_t, depth_s, K = self._r(src), self._r(depth_t), self._r(depth_s), self._r(K)
        _, H, W = src.shape
        T = self._d(self.pose_to_T(pose) if T is None else T).reshape(12)
        rec = np.empty((3, H, W), self.dt); va, pd, cd = (np.empty((H, W), self.dt) for _ in range(3))
        self.lib.orc_warp(H, W, self._p(src), self._p(depth_t), self._p(depth_s), self._p(T), self._p(K),
                          C.c_double(log_scale), self._p(rec), self._p(va), self._p(pd), self._p(cd))
        return rec, va, pd, cd

    def sample_positions(self, depth_t, pose, K, log_scale=0.0):
        """grid_sample's un-normalised sample coordinates (ix, iy) [H,W] of every target pixel under `pose` (diagnostic)"""
        depth_t, K = self._r(depth_t), self._r(K)
        H, W = depth_t.shape
        T = self._d(self.pose_to_T(pose)).reshape(12)
        ix, iy = np.empty((H, W), self.dt), np.empty((H, W), self.dt)
        self.lib.orc_sample_positions(H, W, self._p(depth_t), self._p(T), self._p(K), C.c_double(log_scale), self._p(ix), self._p(iy))
        return ix, iy

    def ssim(self, x, y):
        x, y = self._r(x), self._r(y)
        Cn, H, W = x.shape
        out = np.empty_like(x)
        self.lib.orc_ssim(Cn, H, W, self._p(x), self._p(y), self._p(out))
        return out

    def photometric(self, tgt, src, depth_t, depth_s, pose, K, log_scale=0.0, w_l1=0.15, w_ssim=0.85):
        """compute_photometric_error (helpers.py:8-23) for one pair -> dict of maps."""
        tgt, src, depth_t, depth_s, K = map(self._r, (tgt, src, depth_t, depth_s, K))
        _, H, W = tgt.shape
        T = self._d(self.pose_to_T(pose)).reshape(12)
        o = {k: np.empty((H, W), self.dt) for k in ("diff", "valid", "weight", "auto_err", "auto_mask")}
        o["rec"] = np.empty((3, H, W), self.dt)
        self.lib.orc_photometric(H, W, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s), self._p(T),
                                 self._p(K), C.c_double(log_scale), C.c_double(w_l1), C.c_double(w_ssim),
                                 self._p(o["diff"]), self._p(o["valid"]), self._p(o["weight"]), self._p(o["auto_err"]),
                                 self._p(o["auto_mask"]), self._p(o["rec"]))
        return o

    def cost(self, tgt, src, depth_t, depth_s, pose, K, opts=None, log_scale=0.0):
        opts = opts or default_opts()
        tgt, src, depth_t, depth_s, K = map(self._r, (tgt, src, depth_t, depth_s, K))
        _, H, W = tgt.shape
        T = self._d(self.pose_to_T(pose)).reshape(12)
        return self.lib.orc_cost(H, W, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s), self._p(T),
                                 self._p(K), C.c_double(log_scale), C.byref(opts))

    def linearize(self, tgt, src, depth_t, depth_s, pose, K, opts=None, log_scale=0.0, rows=False, T=None):
        """-> dict(H [np,np], g [np], cost, cost_photo, cost_dc, n_mask [, J1,J2,J3 [H,W,np], E [H,W,3], M [H,W]])."""
        opts = opts or default_opts()
        tgt, src, depth_t, depth_s, K = map(self._r, (tgt, src, depth_t, depth_s, K))
        _, H, W = tgt.shape
        n_p = opts.nparam
        T = self._d(self.pose_to_T(pose) if T is None else T).reshape(12)
        out = LinOut()
        J = [np.empty((H, W, n_p), self.dt) if rows else None for _ in range(3)]
        E = np.empty((H, W, 3), self.dt) if rows else None
        M = np.empty((H, W), self.dt) if rows else None
        self.lib.orc_linearize(H, W, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s), self._p(T),
                               self._p(K), C.c_double(log_scale), C.byref(opts), None, C.byref(out),
                               self._p(J[0]), self._p(J[1]), self._p(J[2]), self._p(E), self._p(M))
        r = dict(H=np.array(out.H[:n_p * n_p]).reshape(n_p, n_p), g=np.array(out.g[:n_p]), cost=out.cost,
                 cost_photo=out.cost_photo, cost_dc=out.cost_dc, n_mask=out.n_mask)
        if rows:
            r.update(J1=J[0], J2=J[1], J3=J[2], E=E, M=M)
        return r

    def linearize_dense(self, tgt, src, depth_t, depth_s, pose, K, opts=None, lambda_depth=0.0, w_prior=0.0, depth0=None):
        """dense mode (pose + per-pixel inverse depth): -> dict(H = Schur complement [6,6], g [6], cost, n_mask,
        g_rho [H,W], D [H,W], B [H,W,6])"""
        opts = opts or default_opts()
        tgt, src, depth_t, depth_s, K = map(self._r, (tgt, src, depth_t, depth_s, K))
        _, H, W = tgt.shape
        T = self._d(self.pose_to_T(pose)).reshape(12)
        out = LinOut()
        gr, D, B = np.zeros((H, W)), np.zeros((H, W)), np.zeros((H, W, 6))
        self.lib.orc_linearize_dense(H, W, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s), self._p(T), self._p(K),
                                     C.byref(opts), None, C.c_double(lambda_depth), C.c_double(w_prior),
                                     self._p(None if depth0 is None else self._r(depth0)), C.byref(out), self._p(gr), self._p(D), self._p(B))
        return dict(H=np.array(out.H[:36]).reshape(6, 6)[:6, :6].copy() if False else np.array([out.H[j * 6 + k] for j in range(6) for k in range(6)]).reshape(6, 6),
                    g=np.array(out.g[:6]), cost=out.cost, n_mask=out.n_mask, g_rho=gr, D=D, B=B)

    def refine_dense(self, tgt, src, depth_t, depth_s, pose, K, opts=None, lambda_depth=1e-2, w_prior=0.0, min_depth=0.06, max_depth=2.67,
                     bits=None, decide=None):
        """-> (pose [6], refined depth [H,W], stats); bits [n_lin,H,W] / decide [n_lin]: replay the engine's decisions"""
        opts = opts or default_opts()
        bits, decide = self._forced(bits, decide)
        tgt, src, depth_s, K = map(self._r, (tgt, src, depth_s, K))
        depth = self._r(depth_t).copy()
        _, H, W = tgt.shape
        pose = self._d(pose).copy()
        stats = np.zeros((opts.n_iters + 1, 4))
        self.lib.orc_refine_dense_forced(H, W, self._p(tgt), self._p(src), self._p(depth), self._p(depth_s), self._p(K), C.byref(opts),
                                         C.c_double(lambda_depth), C.c_double(w_prior), C.c_double(min_depth), C.c_double(max_depth),
                                         self._p(pose), self._p(stats), self._p(bits), self._p(decide))
        return pose, depth, stats

    def refine_dense_window(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, lambda_depth=1.0, w_prior=10.0,
                            min_depth=0.06, max_depth=2.67, bits=None, decide=None):
        """dense window mode: tgt [B,3,H,W], srcs [S,B,3,H,W], depth_t [B,H,W], depth_s [S,B,H,W], poses [2SB,6] ->
        (poses [2SB,6], refined target depth of every directed pair [2SB,H,W], stats [2SB,n_iters+1,4])"""
        opts = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K = map(self._r, (tgt, srcs, depth_t, depth_s, K))
        S, B, _, H, W = srcs.shape
        dt_pairs = np.ascontiguousarray(np.concatenate([np.tile(depth_t, (S, 1, 1)), depth_s.reshape(S * B, H, W)]))   # own target depth
        ds_pairs = np.ascontiguousarray(np.concatenate([depth_s.reshape(S * B, H, W), np.tile(depth_t, (S, 1, 1))]))   # source depth
        pose = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(2 * S * B, 6)).copy()
        stats = np.zeros((2 * S * B, opts.n_iters + 1, 4))
        bits, decide = self._forced(bits, decide)
        self.lib.orc_refine_dense_window_forced(H, W, B, S, self._p(tgt), self._p(srcs), self._p(dt_pairs), self._p(ds_pairs), self._p(K),
                                                C.byref(opts), int(bool(argmin)), C.c_double(lambda_depth), C.c_double(w_prior),
                                                C.c_double(min_depth), C.c_double(max_depth), self._p(pose), self._p(stats),
                                                self._p(bits), self._p(decide))
        return pose, dt_pairs, stats

    def linearize_dense_joint(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, lambda_depth=0.0, w_prior=0.0, depth0=None, rule=1):
        """joint dense mode, ONE target: tgt [3,H,W], srcs [S,3,H,W], depth_t [H,W] (shared by the S forward pairs), depth_s [S,H,W],
        poses [S,6] -> dict(H [6S,6S] reduced system, g [6S], cost, cost_photo, cost_prior, K, share [S], n_mask [S], g_rho [H,W],
        D [H,W], B [H,W,S,6])"""
        opts = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K = map(self._r, (tgt, srcs, depth_t, depth_s, K))
        S, _, H, W = srcs.shape
        pose = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(S, 6))
        NP = 6 * S
        Hm, g, sc = np.zeros((NP, NP)), np.zeros(NP), np.zeros(4 + 2 * S)
        gr, D, B = np.zeros((H, W)), np.zeros((H, W)), np.zeros((H, W, S, 6))
        self.lib.orc_linearize_dense_joint(H, W, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s), self._p(K), C.byref(opts),
                                           int(bool(argmin)), int(rule), self._p(pose), C.c_double(lambda_depth), C.c_double(w_prior),
                                           self._p(None if depth0 is None else self._r(depth0)), self._p(Hm), self._p(g), self._p(sc),
                                           self._p(gr), self._p(D), self._p(B))
        return dict(H=Hm, g=g, cost=sc[0], cost_photo=sc[1], cost_prior=sc[2], K=sc[3], share=sc[4:4 + S], n_mask=sc[4 + S:], g_rho=gr, D=D, B=B)

    def refine_dense_joint(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, lambda_depth=1.0, w_prior=10.0,
                           min_depth=0.06, max_depth=2.67, bits=None, decide=None, rule=0):
        """joint dense mode over the FORWARD group of a window: tgt [B,3,H,W], srcs [S,B,3,H,W], depth_t [B,H,W] (one map per
        target, shared by its S forward pairs), depth_s [S,B,H,W] (fixed), poses [S*B,6] (stacked forward pairs) ->
        (poses [SB,6], refined target depths [B,H,W], stats [SB,n_iters+1,4]); bits [n_lin,SB,H,W] / decide [n_lin,SB]: replay"""
        opts = opts or default_opts()
        tgt, srcs, depth_s, K = map(self._r, (tgt, srcs, depth_s, K))
        S, B, _, H, W = srcs.shape
        depth = self._r(depth_t).copy()
        pose = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(S * B, 6)).copy()
        stats = np.zeros((S * B, opts.n_iters + 1, 4))
        bits, decide = self._forced(bits, decide)
        self.lib.orc_refine_dense_joint(H, W, B, S, self._p(tgt), self._p(srcs), self._p(depth), self._p(depth_s), self._p(K), C.byref(opts),
                                        int(bool(argmin)), int(rule), C.c_double(lambda_depth), C.c_double(w_prior), C.c_double(min_depth),
                                        C.c_double(max_depth), self._p(pose), self._p(stats), self._p(bits), self._p(decide))
        return pose, depth, stats

    # -- dense mode on the reference's own loss (round 4) ------------------------------------------------------------
    def _dref_args(self, tgt, srcs, depth_t, depth_s, K, poses):
        tgt = self._r(tgt); srcs = self._r(srcs); depth_t = self._r(depth_t); depth_s = self._r(depth_s); K = self._r(K)
        S, B = srcs.shape[:2]
        H, W = tgt.shape[-2:]
        assert tgt.shape == (B, 3, H, W) and srcs.shape == (S, B, 3, H, W) and depth_t.shape == (B, H, W) and depth_s.shape == (S, B, H, W)
        poses = self._d(poses).reshape(2 * S * B, 6).copy()
        return tgt, srcs, depth_t.copy(), depth_s, K, poses, S, B, H, W

    def linearize_dense_ref(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, w_init=0.0, depth0=None, min_depth=0.06,
                            max_depth=2.67, lambda_depth=0.0):
        """ONE linearisation of the dense mode on the reference's loss (orc_linearize_dense_ref): tgt [B,3,H,W], srcs [S,B,3,H,W],
        depth_t [B,H,W], depth_s [S,B,H,W], K [B,3,3], poses [2SB,6] (forward pairs source-major, then the inverse pairs) ->
        dict(loss, L_fwd, L_inv, L_dc, L_init, K_f, K_i, g_xi [2SB,6] (w.r.t. the left perturbation), g_rho [B,H,W], H_joint [B,6S,6S],
        g_joint [B,6S], D [B,H,W], B [B,H,W,6S], H_inv [SB,6,6], g_inv [SB,6], g_rho_s [S,B,H,W]: the gradient w.r.t. the SOURCE inverse-depth maps,
        which the engine holds fixed and the reference optimises too)"""
        o = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K, poses, S, B, H, W = self._dref_args(tgt, srcs, depth_t, depth_s, K, poses)
        d0 = None if depth0 is None else self._r(depth0)
        SB, NP, n = S * B, 6 * S, H * W
        scal = np.zeros(7); g_xi = np.zeros((2 * SB, 6)); g_rho = np.zeros((B, H, W)); Hj = np.zeros((B, NP, NP)); gj = np.zeros((B, NP))
        Dq = np.zeros((B, H, W)); Bq = np.zeros((B, H, W, NP)); Hi = np.zeros((SB, 6, 6)); gi = np.zeros((SB, 6))
        g_rho_s = np.zeros((S, B, H, W))
        self.lib.orc_linearize_dense_ref(C.c_int(H), C.c_int(W), C.c_int(B), C.c_int(S), self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                         self._p(d0), self._p(K), C.byref(o), C.c_int(1 if argmin else 0), C.c_double(w_init), C.c_double(min_depth),
                                         C.c_double(max_depth), C.c_double(lambda_depth), self._p(poses), self._p(scal), self._p(g_xi), self._p(g_rho),
                                         self._p(Hj), self._p(gj), self._p(Dq), self._p(Bq), self._p(Hi), self._p(gi), self._p(g_rho_s))
        return dict(loss=scal[0], L_fwd=scal[1], L_inv=scal[2], L_dc=scal[3], L_init=scal[4], K_f=scal[5], K_i=scal[6], g_xi=g_xi, g_rho=g_rho,
                    H_joint=Hj, g_joint=gj, D=Dq, B=Bq, H_inv=Hi, g_inv=gi, g_rho_s=g_rho_s)

    def refine_dense_ref(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06,
                         max_depth=2.67, bits=None):
        """Gauss-Newton on the reference's loss over the poses of all 2SB pairs and the B target inverse-depth maps
        (orc_refine_dense_ref) -> (poses [2SB,6], depth_t [B,H,W], stats [n_iters,7]: loss, L_fwd, L_inv, L_dc, L_init, K_f, K_i)"""
        o = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K, poses, S, B, H, W = self._dref_args(tgt, srcs, depth_t, depth_s, K, poses)
        stats = np.zeros((o.n_iters, 7))
        b, _ = self._forced(bits, None)
        self.lib.orc_refine_dense_ref(C.c_int(H), C.c_int(W), C.c_int(B), C.c_int(S), self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                      self._p(K), C.byref(o), C.c_int(1 if argmin else 0), C.c_double(w_init), C.c_double(lambda_depth),
                                      C.c_double(min_depth), C.c_double(max_depth), self._p(poses), self._p(stats), self._p(b))
        return poses, depth_t, stats

    def refine_dense_ref_free(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06,
                              max_depth=2.67, bits=None):
        """orc_refine_dense_ref_free: as refine_dense_ref with the SOURCE depth maps as unknowns too (the reference's optimize_depth_pred
        optimises them, with no prior): every inverse pair is a group of its pose and the source map it back-projects
        -> (poses [2SB,6], depth_t [B,H,W], depth_s [S,B,H,W], stats [n_iters,7])"""
        o = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K, poses, S, B, H, W = self._dref_args(tgt, srcs, depth_t, depth_s, K, poses)
        depth_s = np.ascontiguousarray(depth_s).copy()
        stats = np.zeros((o.n_iters, 7))
        b, _ = self._forced(bits, None)
        self.lib.orc_refine_dense_ref_free(C.c_int(H), C.c_int(W), C.c_int(B), C.c_int(S), self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                           self._p(K), C.byref(o), C.c_int(1 if argmin else 0), C.c_double(w_init), C.c_double(lambda_depth),
                                           C.c_double(min_depth), C.c_double(max_depth), self._p(poses), self._p(stats), self._p(b))
        return poses, depth_t, depth_s, stats

    def refine_dense_ref_q(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06,
                           max_depth=2.67, bits=None):
        """the same in the reference's parametrisation (optimizer.py:194-198, 235-239: QUARTER-resolution map, upsampled x4 bilinear every
        linearisation; orc_refine_dense_ref_q) -> (poses, depth_t [B,H,W] = the upsampled map, stats, rho_q [B,H/4,W/4])"""
        o = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K, poses, S, B, H, W = self._dref_args(tgt, srcs, depth_t, depth_s, K, poses)
        assert H % 4 == 0 and W % 4 == 0
        stats = np.zeros((o.n_iters, 7))
        rq = np.zeros((B, H // 4, W // 4))
        b, _ = self._forced(bits, None)
        self.lib.orc_refine_dense_ref_q(C.c_int(H), C.c_int(W), C.c_int(B), C.c_int(S), self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                        self._p(K), C.byref(o), C.c_int(1 if argmin else 0), C.c_double(w_init), C.c_double(lambda_depth),
                                        C.c_double(min_depth), C.c_double(max_depth), self._p(poses), self._p(stats), self._p(b), self._p(rq))
        return poses, depth_t, stats, rq

    def refine_dense_ref_q_free(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06,
                                max_depth=2.67, bits=None):
        """orc_refine_dense_ref_q_free: the reference's complete leaf set -- the QUARTER-resolution maps of the target and of every source are
        unknowns (optimizer.py:194-198) -> (poses, depth_t [B,H,W], depth_s [S,B,H,W] (both upsampled), stats)"""
        o = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K, poses, S, B, H, W = self._dref_args(tgt, srcs, depth_t, depth_s, K, poses)
        assert H % 4 == 0 and W % 4 == 0
        depth_s = np.ascontiguousarray(depth_s).copy()
        stats = np.zeros((o.n_iters, 7))
        b, _ = self._forced(bits, None)
        self.lib.orc_refine_dense_ref_q_free(C.c_int(H), C.c_int(W), C.c_int(B), C.c_int(S), self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                             self._p(K), C.byref(o), C.c_int(1 if argmin else 0), C.c_double(w_init), C.c_double(lambda_depth),
                                             C.c_double(min_depth), C.c_double(max_depth), self._p(poses), self._p(stats), self._p(b), self._p(None))
        return poses, depth_t, depth_s, stats

    def up4(self, q):
        """F.interpolate(q, x4, bilinear, align_corners=False) of one map [h,w] -> [4h,4w] (orc_up4)"""
        q = np.ascontiguousarray(q, np.float64)
        out = np.empty((4 * q.shape[0], 4 * q.shape[1]))
        self.lib.orc_up4(C.c_int(out.shape[0]), C.c_int(out.shape[1]), self._p(q), self._p(out))
        return out

    def down4(self, full):
        """F.interpolate(full, (H/4, W/4), bilinear, align_corners=False) of one map (orc_down4)"""
        full = np.ascontiguousarray(full, np.float64)
        out = np.empty((full.shape[0] // 4, full.shape[1] // 4))
        self.lib.orc_down4(C.c_int(full.shape[0]), C.c_int(full.shape[1]), self._p(full), self._p(out))
        return out

    def up4_adjoint(self, full):
        """transpose of up4: [H,W] -> [H/4,W/4] (the chain rule from d / d full-resolution map to d / d quarter-resolution map)"""
        full = np.ascontiguousarray(full, np.float64)
        out = np.empty((full.shape[0] // 4, full.shape[1] // 4))
        self.lib.orc_up4_adjoint(C.c_int(full.shape[0]), C.c_int(full.shape[1]), self._p(full), self._p(out))
        return out

    def ground_height(self, depth, K):
        """DNet camera-height map and ground mask of one image (dnet_layers.py:259-304,319-322)"""
        depth, K = self._r(depth), self._r(K)
        H, W = depth.shape
        h, m = np.empty((H, W), self.dt), np.empty((H, W), self.dt)
        self.lib.orc_ground_height(H, W, self._p(depth), self._p(K), self._p(h), self._p(m))
        return h, m

    def scale_recovery(self, depth, K, real_cam_height):
        """ScaleRecovery.forward for a batch: depth [B,H,W], K [B,3,3] -> (scale, median camera height)"""
        depth, K = self._r(depth), self._r(K)
        B, H, W = depth.shape
        med = C.c_double()
        s = self.lib.orc_scale_recovery(B, H, W, self._p(depth), self._p(K), C.c_double(real_cam_height), C.byref(med))
        return s, med.value

    def window_select(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None):
        """per-pixel min-over-sources selection masks of the forward pairs at the given poses:
        tgt [B,3,H,W], srcs [S,B,3,H,W], depth_t [B,H,W], depth_s [S,B,H,W], K [B,3,3], poses [S*B,6] -> masks [S*B,H,W]"""
        opts = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K = map(self._r, (tgt, srcs, depth_t, depth_s, K))
        S, B, _, H, W = srcs.shape
        T = np.ascontiguousarray(np.stack([self._d(self.pose_to_T(p)).reshape(12) for p in np.asarray(poses).reshape(-1, 6)]))
        mask = np.empty((S * B, H, W), self.dt)
        self.lib.orc_window_select(H, W, B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s), self._p(K),
                                   C.byref(opts), self._p(T), None, self._p(mask))
        return mask

    def refine_window(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, log_scale=None, bits=None, decide=None, rule=0):
        """window mode (forward + inverse pairs, optional min-over-sources selection): poses [2*S*B,6] in the stacked
        order of train_mono.py:54-62 -> (poses [2SB,6], log_scale [2SB] or None, stats [2SB,n_iters+1,4]).
        rule 0: every pair its own problem; 1: the reference's compute_optimization_loss (optimizer.py:47-86) is the scalar minimised"""
        opts = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K = map(self._r, (tgt, srcs, depth_t, depth_s, K))
        S, B, _, H, W = srcs.shape
        pose = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(2 * S * B, 6)).copy()
        ls = None if log_scale is None else np.ascontiguousarray(np.asarray(log_scale, dtype=np.float64)).copy()
        stats = np.zeros((2 * S * B, opts.n_iters + 1, 4))
        bits, decide = self._forced(bits, decide)
        self.lib.orc_refine_window_rule(H, W, B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s), self._p(K),
                                        C.byref(opts), int(bool(argmin)), int(rule), self._p(pose), self._p(ls), self._p(stats),
                                        self._p(bits), self._p(decide), None)
        return pose, ls, stats

    def linearize_window(self, tgt, srcs, depth_t, depth_s, K, poses, opts=None, argmin=True, rule=0, log_scale=None):
        """one linearisation of a whole window at `poses` [2SB,6] -> dict(H [2SB,np,np], g [2SB,np], cost, cost_photo, cost_dc, n_mask [2SB])"""
        opts = opts or default_opts()
        tgt, srcs, depth_t, depth_s, K = map(self._r, (tgt, srcs, depth_t, depth_s, K))
        S, B, _, H, W = srcs.shape
        N, n_p = 2 * S * B, opts.nparam
        pose = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(N, 6))
        ls = None if log_scale is None else np.ascontiguousarray(np.asarray(log_scale, dtype=np.float64))
        out = (LinOut * N)()
        self.lib.orc_linearize_window(H, W, B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s), self._p(K),
                                      C.byref(opts), int(bool(argmin)), int(rule), self._p(pose), self._p(ls), out)
        return dict(H=np.stack([np.array(o.H[:n_p * n_p]).reshape(n_p, n_p) for o in out]), g=np.stack([np.array(o.g[:n_p]) for o in out]),
                    cost=np.array([o.cost for o in out]), cost_photo=np.array([o.cost_photo for o in out]),
                    cost_dc=np.array([o.cost_dc for o in out]), n_mask=np.array([o.n_mask for o in out]))

    def flip_stats_reset(self):
        self.lib.orc_flip_stats_reset()

    def flip_stats(self, n_lin):
        """flip statistics of the forced replays since the last reset (this thread): per linearisation (pixels the oracle would
        have masked differently from the engine, those of them that are not near-ties) -> (n [n_lin], hard [n_lin])"""
        n = (C.c_long * 64)(); h = (C.c_long * 64)()
        self.lib.orc_flip_stats(n, h, 64)
        return np.array(n[:n_lin]), np.array(h[:n_lin])

    def refine(self, tgt, src, depth_t, depth_s, pose, K, opts=None, log_scale=0.0, bits=None, decide=None):
        """GN/LM refinement of one directed pair -> (pose [6], log_scale, stats [n_iters+1,4]).
        bits [n_lin,H,W] uint16 / decide [n_lin] int32 (n_lin = n_iters, +1 for LM): replay the engine's per-pixel mask / validity
        and accept decisions instead of taking them here (tie-proof parity, see g_force_bits in tcsfm_oracle.c)."""
        opts = opts or default_opts()
        tgt, src, depth_t, depth_s, K = map(self._r, (tgt, src, depth_t, depth_s, K))
        _, H, W = tgt.shape
        pose = self._d(pose).copy()
        ls = C.c_double(log_scale)
        stats = np.zeros((opts.n_iters + 1, 4))
        bits, decide = self._forced(bits, decide)
        self.lib.orc_refine_forced(H, W, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s), self._p(K),
                                   C.byref(opts), self._p(pose), C.byref(ls), self._p(stats), self._p(bits), self._p(decide))
        return pose, ls.value, stats

    def refine_record(self, tgt, src, depth_t, depth_s, pose, K, opts=None, log_scale=0.0):
        """free-running refine that also returns its own decision trace -> (pose, log_scale, stats, bits [n_lin,H,W], decide [n_lin])"""
        opts = opts or default_opts()
        tgt, src, depth_t, depth_s, K = map(self._r, (tgt, src, depth_t, depth_s, K))
        _, H, W = tgt.shape
        pose = self._d(pose).copy()
        ls = C.c_double(log_scale)
        n_lin = opts.n_iters + (1 if opts.solver == 1 and opts.n_iters > 0 else 0)
        stats = np.zeros((opts.n_iters + 1, 4))
        bits = np.zeros((n_lin, H, W), np.uint16); decide = np.zeros(n_lin, np.int32)
        self.lib.orc_refine_record(H, W, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s), self._p(K),
                                   C.byref(opts), self._p(pose), C.byref(ls), self._p(stats), self._p(bits), self._p(decide))
        return pose, ls.value, stats, bits, decide
