#!/usr/bin/env python3
"""Diagnostic (appendix R4): the ONE source-map pixel of the 192x640, S = 2 free-source run that deviates from the oracle's twin by more than
1e-4 after three iterations.  On the GPU box: runs the engine with decision recording and saves its bits and maps
(gpurun_out/free_pixel.npz); here, with that file: re-runs the float64 oracle for 2 and 3 iterations under the replayed decisions and prints
the pixel's neighbourhood -- per-channel SSIM values before the clamp and depth-consistency ratios of the inverse pair at the state before
the third linearisation -- to tell which switch sits next to it.
    GPU box:   python scripts/diag/free_source_pixel.py gpu
    here:      python scripts/diag/free_source_pixel.py cpu"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out", "free_pixel.npz")
H, W, S, B, mind, maxd, n_it = 192, 640, 2, 1, 0.06, 2.67, 3
N = 2 * S * B

def window():
    import test_gpu_dense_reference as T
    return T._window(B, S, H, W, seed=31)

if sys.argv[1] == "gpu":
    import torch
    import test_gpu_dense_reference as T
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    w = window()
    t = {k: T._dev(v) for k, v in w.items()}
    e = Engine(H, W, N)
    o = default_opts(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0, free_source_depths=1)
    e.trace_begin(n_it, N)
    pose, depth, st = e.refine_dense_window(t["tgt"], t["srcs"], t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous(), t["K"], t["pose"], o, stats=True, argmin=True)
    bits, _ = e.trace_end()
    np.savez_compressed(OUT, bits=np.asarray(bits).reshape(n_it, N, H * W), depth=depth.cpu().numpy(), pose=pose.cpu().numpy())
    print("saved", OUT)
else:
    from oracle.oracle import Oracle, default_opts as oopts
    z = np.load(OUT)
    w = window()
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    orc = Oracle("f64")
    args = (f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]))
    bits = z["bits"].astype(np.uint16)
    p3, d3, ds3, _ = orc.refine_dense_ref_free(*args, oopts(n_iters=3, w_dc=0.15), argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=mind, max_depth=maxd, bits=bits)
    p2, d2, ds2, _ = orc.refine_dense_ref_free(*args, oopts(n_iters=2, w_dc=0.15), argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=mind, max_depth=maxd, bits=bits[:2])
    gpu_src = z["depth"][S * B:, 0].reshape(S, B, H, W).astype(np.float64)
    dev = np.abs(gpu_src / ds3 - 1)
    s_, b_, y, x = np.unravel_index(np.argmax(dev), dev.shape)
    print("worst pixel", (s_, b_, y, x), "deviation", dev[s_, b_, y, x], "step of iteration 3 (oracle):", ds3[s_, b_, y, x] / ds2[s_, b_, y, x] - 1, " (gpu):", gpu_src[s_, b_, y, x] / ds2[s_, b_, y, x] - 1)
    # the inverse pair of source s_ at the state before the third linearisation: target image = the source image, its depth = the source map
    m = s_ * B + b_
    T12 = np.zeros(12); orc.lib.orc_pose_to_T(p2[S * B + m].ctypes.data_as(C.c_void_p), T12.ctypes.data_as(C.c_void_p))
    tgt_i = np.ascontiguousarray(args[1][s_, b_]); src_i = np.ascontiguousarray(args[0][b_]); dt_i = np.ascontiguousarray(ds2[s_, b_]); ds_i = np.ascontiguousarray(d2[b_]); K = np.ascontiguousarray(args[4][b_])
    out = np.zeros(16)
    print("inverse pair, 5x5 neighbourhood: [dd ratio (cd-pd)/(cd+pd), valid, per channel (rec - tgt, SSIM raw before the clamp)]")
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            orc.lib.orc_pixel_debug(C.c_int(H), C.c_int(W), tgt_i.ctypes.data_as(C.c_void_p), src_i.ctypes.data_as(C.c_void_p), dt_i.ctypes.data_as(C.c_void_p), ds_i.ctypes.data_as(C.c_void_p),
                                    T12.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p), C.c_int(int(x + dx)), C.c_int(int(y + dy)), out.ctypes.data_as(C.c_void_p))
            flag = [("SSIMclamp" if (out[6 + 2 * c] < 2e-4 or out[6 + 2 * c] > 1 - 2e-4) else "") for c in range(3)]
            print((dy, dx), np.round(out[3], 6), int(out[4]), [(round(out[5 + 2 * c], 5), round(out[6 + 2 * c], 6)) for c in range(3)], [f for f in flag if f])
    # the tail of the deviations: isolated outlier or the end of a continuous (conditioning) tail?
    flat = np.argsort(dev.ravel())[::-1][:12]
    print("largest deviations: (s, b, y, x), deviation, oracle step of iteration 3, gpu step")
    for f in flat:
        i = np.unravel_index(f, dev.shape)
        print(tuple(int(v) for v in i), f"{dev[i]:.2e}", f"{ds3[i] / ds2[i] - 1:+.5f}", f"{gpu_src[i] / ds2[i] - 1:+.5f}", "bits it3 inverse pair:", hex(int(bits[2, S * B + i[0] * B + i[1], i[2] * W + i[3]])),
              " forward:", hex(int(bits[2, i[0] * B + i[1], i[2] * W + i[3]])))
    print("quantiles of the deviation 0.5 / 0.99 / 0.9999 / 0.99999:", [float(np.quantile(dev, q)) for q in (0.5, 0.99, 0.9999, 0.99999)])
    print("|step| of iteration 3 at the worst pixel vs the map's median |step|:", abs(ds3[s_, b_, y, x] / ds2[s_, b_, y, x] - 1), float(np.median(np.abs(ds3 / ds2 - 1))))
    # the 2x2 cluster is one bilinear cell: look for the forward pair's samples that land in it (their adjoint is scattered to its four pixels)
    mf = s_ * B + b_
    orc.lib.orc_pose_to_T(p2[mf].ctypes.data_as(C.c_void_p), T12.ctypes.data_as(C.c_void_p))
    tgt_f = np.ascontiguousarray(args[0][b_]); src_f = np.ascontiguousarray(args[1][s_, b_]); dt_f = np.ascontiguousarray(d2[b_]); ds_f = np.ascontiguousarray(ds2[s_, b_])
    print("forward pair: target pixels whose sample falls into the cell rows %d..%d, columns %d..%d: (v, u), ix, iy, (cd-pd)/(cd+pd), valid, per channel (rec - tgt, raw SSIM), bits" % (y, y + 1, x - 1, x))
    for vv in range(max(0, y - 20), min(H, y + 9)):             # (the flow here is about (+20, +5) pixels)
        for uu in range(max(0, x - 45), min(W, x + 6)):
            orc.lib.orc_pixel_debug(C.c_int(H), C.c_int(W), tgt_f.ctypes.data_as(C.c_void_p), src_f.ctypes.data_as(C.c_void_p), dt_f.ctypes.data_as(C.c_void_p), ds_f.ctypes.data_as(C.c_void_p),
                                    T12.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p), C.c_int(uu), C.c_int(vv), out.ctypes.data_as(C.c_void_p))
            if vv == y and uu == x: print('  (the sample of the pixel itself lands at', round(out[0], 3), round(out[1], 3), ')')
            if x - 2 <= out[0] <= x + 1 and y - 1 <= out[1] <= y + 2:
                print((vv, uu), round(out[0], 4), round(out[1], 4), f"{out[3]:+.3e}", int(out[4]), [(round(out[5 + 2 * c], 5), round(out[6 + 2 * c], 6)) for c in range(3)], hex(int(bits[2, mf, vv * W + uu])))
