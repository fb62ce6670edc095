"""diagnostic: per-pixel depth disagreements of the dense mode against the replayed float64 oracle -- where do they start and what
discontinuity of the residual's derivative does the pixel (or one of its 3x3 neighbours) sit on?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from oracle.oracle import Oracle, default_opts as oopts
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 320)
N = 2
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
b = synth.make_batch(N, H, W, seed0=8, both_directions=True)
v, u = np.mgrid[0:H, 0:W]
d0 = (b["depth_t"] * (1 + 0.03 * np.sin(u / 23.0) * np.cos(v / 17.0))[None, None]).astype(np.float32)
orc = Oracle("f64")
kw = dict(min_depth=0.03, max_depth=3.0)
e = Engine(H, W, N)
prev_bad = {}
for iters in (1, 2, 3, 4):
    o = default_opts(n_iters=iters, lambda_depth=1.0, prior_depth=10.0, **kw)
    e.trace_begin(iters, N)
    pose, depth, st = e.refine_dense(t(b["tgt"]), t(b["src"]), t(d0), t(b["depth_s"]), t(b["K"]), t(b["pose_init"]), o, stats=True)
    bits, dec = e.trace_end()
    depth = depth.cpu().numpy()[:, 0]
    for n in range(N):
        rp, rd, rst = orc.refine_dense(b["tgt"][n], b["src"][n], d0[n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n], oopts(n_iters=iters),
                                       lambda_depth=1.0, w_prior=10.0, bits=bits[:, n], decide=dec[:, n], **kw)
        rel = np.abs(depth[n] / rd - 1)
        bad = np.argwhere(rel > 3e-5)
        print(f"iters {iters} pair {n}: max rel {rel.max():.3e}  n(>3e-5) {len(bad)}  q99.99 {np.quantile(rel, 0.9999):.2e}")
        for (y, x) in bad[:6]:
            new = (n, y, x) not in prev_bad
            print(f"   px ({y},{x}) rel {rel[y, x]:.3e} {'NEW' if new else ''} bits {[int(bits[i, n, y, x]) for i in range(iters)]}")
            if new:
                prev_bad[(n, y, x)] = iters
                # oracle state at the START of the iteration where the disagreement appeared: replay iters-1 iterations
                if iters > 1:
                    pp, dd, _ = orc.refine_dense(b["tgt"][n], b["src"][n], d0[n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n], oopts(n_iters=iters - 1),
                                                 lambda_depth=1.0, w_prior=10.0, bits=bits[:iters - 1, n], decide=dec[:iters - 1, n], **kw)
                else:
                    pp, dd = b["pose_init"][n].astype(np.float64), d0[n, 0]
                T = orc.pose_to_T(pp).reshape(12)
                lin = orc.linearize_dense(b["tgt"][n], b["src"][n], dd, b["depth_s"][n, 0], pp, b["K"][n], oopts(), lambda_depth=1.0, w_prior=10.0, depth0=d0[n, 0])
                print(f"      oracle: depth {dd[y, x]:.6f} (range {1/3.0:.3f}..{1/0.03:.1f}) g_rho {lin['g_rho'][y, x]:+.3e} D {lin['D'][y, x]:.3e} step {-lin['g_rho'][y, x] / (2 * lin['D'][y, x] + 1e-300):+.3e} rho {1 / dd[y, x]:.4f}")
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        yy, xx = min(max(y + dy, 0), H - 1), min(max(x + dx, 0), W - 1)
                        out = np.zeros(16)
                        orc.lib.orc_pixel_debug(H, W, orc._p(orc._r(b["tgt"][n])), orc._p(orc._r(b["src"][n])), orc._p(orc._r(dd)), orc._p(orc._r(b["depth_s"][n, 0])),
                                                orc._p(orc._d(T)), orc._p(orc._r(b["K"][n])), int(xx), int(yy), orc._p(out))
                        fx, fy = out[0] - np.round(out[0]), out[1] - np.round(out[1])
                        flags = []
                        if abs(fx) < 1e-5 or abs(fy) < 1e-5: flags.append(f"CELL-TIE fx {fx:.1e} fy {fy:.1e}")
                        if abs(out[3]) < 1e-6: flags.append(f"DC-SIGN-TIE rel dif {out[3]:.1e}")
                        for ch in range(3):
                            if abs(out[5 + 2 * ch]) < 1e-6: flags.append(f"L1-SIGN-TIE ch{ch} r {out[5 + 2 * ch]:.1e}")
                            if abs(out[6 + 2 * ch]) < 1e-6 or abs(out[6 + 2 * ch] - 1) < 1e-6: flags.append(f"SSIM-CLAMP-TIE ch{ch} raw {out[6 + 2 * ch]:.2e}")
                        if flags or (dy == 0 and dx == 0):
                            print(f"      nb ({dy:+d},{dx:+d}) ix {out[0]:.6f} iy {out[1]:.6f} dif/sum {out[3]:+.2e} r {out[5]:+.1e},{out[7]:+.1e},{out[9]:+.1e} raw {out[6]:+.2e},{out[8]:+.2e},{out[10]:+.2e} {flags}")
