"""Coarse-to-fine study on the float64 oracle: does a 3-level pyramid widen the convergence basin to SURVEY
8d's initialisation (GT + N(0, 0.01^2) on translation, N(0, 0.002^2) on rotation)?

Pyramid level l: images and INVERSE depths area-averaged over 2^l x 2^l blocks, intrinsics scaled (fx, fy, cx + 1/2, cy + 1/2
divided by 2^l, minus 1/2: pixel centres).  LM at every level, coarse to fine.  'Converged' = the final pose is within
`tol` (relative translation / rotation error) of the full-resolution optimum (a long LM run from the ground truth).
    python scripts/experiments/pyramid_study.py [n_seeds] [H] [W]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.oracle import Oracle, default_opts
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (192, 640)
orc = Oracle("f64")


def down(img, f):
    c = img.reshape(img.shape[:-2] + (img.shape[-2] // f, f, img.shape[-1] // f, f))
    return c.mean(axis=(-3, -1))


def level(p, l):
    f = 2 ** l
    K = p["K"].astype(np.float64).copy()
    K[0, 0] /= f; K[1, 1] /= f
    K[0, 2] = (K[0, 2] + 0.5) / f - 0.5; K[1, 2] = (K[1, 2] + 0.5) / f - 0.5
    inv = lambda d: 1.0 / down(1.0 / d.astype(np.float64), f)
    return dict(tgt=down(p["tgt"].astype(np.float64), f), src=down(p["src"].astype(np.float64), f), depth_t=inv(p["depth_t"]), depth_s=inv(p["depth_s"]), K=K)


def run(p, pose, schedule):
    """schedule: list of (level, iterations) coarse to fine, LM"""
    for l, its in schedule:
        q = level(p, l) if l > 0 else dict(tgt=p["tgt"], src=p["src"], depth_t=p["depth_t"], depth_s=p["depth_s"], K=p["K"])
        pose, _, st = orc.refine(q["tgt"], q["src"], q["depth_t"], q["depth_s"], pose, q["K"], default_opts(n_iters=its, solver=1, lambda0=1e-3))
    return pose


def err(a, b):
    """(translation error relative to the motion, rotation error in rad)"""
    return np.linalg.norm(a[:3] - b[:3]) / np.linalg.norm(b[:3]), np.linalg.norm(a[3:] - b[3:])


schedules = {"single scale, 4 LM": [(0, 4)], "single scale, 16 LM": [(0, 16)], "3 levels 6/4/4": [(2, 6), (1, 4), (0, 4)],
             "3 levels 10/6/8": [(2, 10), (1, 6), (0, 8)], "4 levels 10/6/4/8": [(3, 10), (2, 6), (1, 4), (0, 8)],
             "4 levels 10/8/8/16": [(3, 10), (2, 8), (1, 8), (0, 16)]}


def one_seed(seed):
    p = synth.make_pair(H, W, seed=seed, dtype=np.float32)
    opt = run(p, p["pose_gt"].astype(np.float64), [(0, 40)])                      # the photometric optimum
    rows = []
    for sig_t, sig_r, tag in ((0.001, 0.0003, "posenet"), (0.01, 0.002, "survey8d")):
        init = synth.perturb_pose(p["pose_gt"].astype(np.float64), seed, sigma_t=sig_t, sigma_r=sig_r)
        for name, sch in schedules.items():
            out = run(p, init.copy(), sch)
            rows.append((name, tag, err(out, opt), err(init, opt)))
    return rows


if __name__ == "__main__":
    import multiprocessing as mp
    t0 = time.time()
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        allrows = pool.map(one_seed, range(nseeds))
    print(f"{nseeds} seeds in {time.time() - t0:.0f} s", flush=True)
    summary = {}
    for name in schedules:
        for tag in ("posenet", "survey8d"):
            e = np.array([r[2] for rows in allrows for r in rows if r[0] == name and r[1] == tag])
            e0 = np.array([r[3] for rows in allrows for r in rows if r[0] == name and r[1] == tag])
            # converged: translation within 2 % of the motion and rotation within 2e-4 rad (0.011 deg) of the photometric optimum
            conv = (e[:, 0] < 2e-2) & (e[:, 1] < 2e-4)
            summary[f"{name} | {tag}"] = {"converged": float(np.mean(conv)), "median_trans_err_rel": float(np.median(e[:, 0])), "median_rot_err_rad": float(np.median(e[:, 1])),
                                          "initial_trans_err_rel": float(np.median(e0[:, 0])), "initial_rot_err_rad": float(np.median(e0[:, 1])), "n": int(len(e))}
    print(json.dumps(summary, indent=1))
    json.dump(summary, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", f"r02_pyramid_study_{H}x{W}.json"), "w"), indent=1)
