"""PoseNet timings at 640x192: the library's gfx950 convolution stack vs the same network in PyTorch (MIOpen convolutions, torch
GroupNorm) on the same GPU -- a single forward on the fwd + inv pair of one window, and the whole coupled loop of
solve_pose_iteratively (4 iterations: 4 network evaluations + 3 warps)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import standins
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine
from tightly_coupled_sfm_amd.posenet import PoseNetHIP

H, W = 192, 640
def timeit(f, n=200, warm=20):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for B, S in ((1, 1), (1, 2), (8, 1)):
    N = 2 * B * S
    w = standins.make_window(B, S, H, W, seed0=90)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    from oracle.oracle import Oracle
    o64 = Oracle("f64")
    dt = t(o64.disp_to_depth(w["disp_t"], 0.06, 2.67)[1]); ds = t(o64.disp_to_depth(w["disp_s"], 0.06, 2.67)[1])
    tg, sr, K = t(w["target"]), t(w["sources"]), t(w["K"])
    sd = standins.posenet_params(0)
    e = Engine(H, W, N)
    net = PoseNetHIP(e, N, sd)
    twin = standins.PoseNetTwin(sd).cuda().eval()
    T = tg.repeat(S, 1, 1, 1); Sx = sr.reshape(S * B, 3, H, W)
    imgs = torch.cat([torch.cat([T, Sx], 1), torch.cat([Sx, T], 1)]).contiguous()
    Dt = dt.repeat(S, 1, 1, 1); Ds = ds.reshape(S * B, 1, H, W)
    d_t = torch.cat([Dt, Ds]).contiguous(); d_s = torch.cat([Ds, Dt]).contiguous(); Kk = K.repeat(2 * S, 1, 1).contiguous()
    tgt, src = imgs[:, :3].contiguous(), imgs[:, 3:].contiguous()
    def torch_loop():
        with torch.no_grad():
            full = twin(imgs)
            for _ in range(3):
                full = full + twin(e.posenet_input(tgt, src, d_t, d_s, full.contiguous(), Kk))
        return full
    with torch.no_grad():
        us_fwd_hip = timeit(lambda: net(imgs)); us_fwd_torch = timeit(lambda: twin(imgs))
    us_loop_hip = timeit(lambda: net.solve_pose_iteratively(4, tg, sr, dt, ds, K), n=100)
    us_loop_torch = timeit(torch_loop, n=100)
    print(json.dumps({"window": f"B={B} S={S} ({N} six-channel images of {W}x{H})", "forward_us": {"hip": round(us_fwd_hip, 1), "torch_miopen": round(us_fwd_torch, 1)},
                      "coupled_loop_4_iterations_us": {"hip_in_library": round(us_loop_hip, 1), "torch_network_plus_library_warp": round(us_loop_torch, 1)},
                      "GFLOP_per_forward": round(0.745 * N, 2)}), flush=True)
