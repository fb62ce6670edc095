"""GPU: the HIP PoseNet (csrc/posenet_kernel.h: fp32 matrix-core convolutions, weight standardisation folded into the loaded
weights, GroupNorm + ReLU applied by the consumer) and the in-library coupled pose loop, against the golden G12 produced by the
reference's own pose_model / solve_pose_iteratively, and against a plain-PyTorch fp32 twin on other sizes."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _window(B, S, H, W):
    import standins
    from oracle.oracle import Oracle
    w = standins.make_window(B, S, H, W, seed0=90)
    o64 = Oracle("f64")
    w["depth_t"] = o64.disp_to_depth(w["disp_t"], 0.06, 2.67)[1].astype(np.float32)
    w["depth_s"] = o64.disp_to_depth(w["disp_s"], 0.06, 2.67)[1].astype(np.float32)
    return w


@pytest.mark.parametrize("tag,H,W,N", [("a", 48, 160, 4), ("b", 192, 640, 2)])
def test_posenet_forward_vs_reference_golden(tag, H, W, N):
    """pose_model(imgs) (pose_models.py:122-137) with seeded parameters: 1e-5 relative on the poses, features of the last layer"""
    import standins
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine
    from tightly_coupled_sfm_amd.posenet import PoseNetHIP
    g = load_golden("posenet")
    b = synth.make_batch(N, H, W, seed0=40, both_directions=True)
    x = np.concatenate([b["tgt"], b["src"]], 1)
    assert np.allclose([float(x.astype(np.float64).sum()), float(np.abs(x).max())], g[f"{tag}_in_checksum"], rtol=0, atol=1e-6)
    e = Engine(H, W, N)
    net = PoseNetHIP(e, N, standins.posenet_params(int(g["seed"])))
    pose = net(_t(x)).cpu().numpy()
    ref = g[f"{tag}_pose"]
    assert np.max(np.abs(pose - ref)) < 1e-5 * np.abs(ref).max(), (np.max(np.abs(pose - ref)) / np.abs(ref).max())
    assert np.max(np.abs(pose - ref) / np.maximum(np.abs(ref), 1e-3)) < 1e-4       # and element-wise


@pytest.mark.parametrize("H,W,N", [(64, 96, 3), (100, 333, 1), (192, 640, 5)])
def test_posenet_forward_vs_torch_twin(H, W, N):
    """odd sizes (ragged tiles, odd output extents) and a batch that is not a power of two, against the plain-PyTorch fp32 twin
    run on the same GPU"""
    import standins
    from tightly_coupled_sfm_amd.engine import Engine
    from tightly_coupled_sfm_amd.posenet import PoseNetHIP, is_reference_posenet
    rng = np.random.default_rng(H * W + N)
    x = _t(rng.uniform(0, 1, size=(N, 6, H, W)))
    sd = standins.posenet_params(3)
    twin = standins.PoseNetTwin(sd).cuda().eval()
    assert is_reference_posenet(twin)
    with torch.no_grad():
        ref = twin(x).cpu().numpy()
    net = PoseNetHIP(Engine(H, W, N), N, twin)                      # parameters taken from the module
    pose = net(x).cpu().numpy()
    assert np.max(np.abs(pose - ref)) < 1e-5 * np.abs(ref).max(), np.max(np.abs(pose - ref)) / np.abs(ref).max()
    again = net(x).cpu().numpy()
    assert np.array_equal(again, pose)                              # deterministic (fixed-order reductions, no atomics)
    # batch independent: bit for bit among batches in the same work-split regime (up to 4 images / more: csrc tcsfm_posenet_create),
    # to rounding (the K split changes the summation order) across the two
    M = 3 if N <= 4 else 7
    more = PoseNetHIP(Engine(H, W, M), M, sd)(torch.cat([x[N - 1:N], _t(rng.uniform(0, 1, size=(M - 1, 6, H, W)))]).contiguous()).cpu().numpy()
    assert np.array_equal(more[0], pose[N - 1])
    one = PoseNetHIP(Engine(H, W, 1), 1, sd)(x[N - 1:N].contiguous()).cpu().numpy()
    if N <= 4:
        assert np.array_equal(one[0], pose[N - 1])
    else:
        assert np.max(np.abs(one[0] - pose[N - 1])) < 3e-6 * np.abs(pose).max()


def test_coupled_pose_loop_vs_reference_golden():
    """solve_pose_iteratively (train_mono.py:41-81) with the reference's PoseNet in the loop, entirely inside the library:
    B=2 targets x S=2 sources, 4 iterations -- the reference's stacked poses of all 8 directed pairs"""
    import standins
    from tightly_coupled_sfm_amd.engine import Engine
    from tightly_coupled_sfm_amd.posenet import PoseNetHIP
    g = load_golden("posenet")
    B, S, H, W = 2, 2, 48, 160
    w = _window(B, S, H, W)
    e = Engine(H, W, 2 * S * B)
    net = PoseNetHIP(e, 2 * S * B, standins.posenet_params(int(g["seed"])))
    poses, stacked = net.solve_pose_iteratively(4, _t(w["target"]), _t(w["sources"]), _t(w["depth_t"]), _t(w["depth_s"]), _t(w["K"]))
    poses, stacked = poses.cpu().numpy(), stacked.cpu().numpy()
    ref = g["loop_stacked"]
    assert stacked.shape == ref.shape == (2 * S * B, 4, 6)
    for it in range(4):     # the loop feeds warped images back into the network: rounding differences grow slowly with the iterate
        err = np.max(np.abs(stacked[:, it] - ref[:, it])) / np.abs(ref[:, it]).max()
        assert err < (1e-5 if it == 0 else 2e-5), (it, err)
    assert np.array_equal(poses, stacked[:, -1]) and np.max(np.abs(poses - g["loop_poses"])) < 2e-5 * np.abs(g["loop_poses"]).max()


def test_solve_pose_iteratively_drop_in_uses_the_library_network():
    """train_mono.solve_pose_iteratively handed a module with the reference PoseNet's parameters runs network AND warps in the
    library (no torch convolution is executed) and returns the reference's structure"""
    import standins
    from tightly_coupled_sfm_amd import train_mono
    g = load_golden("posenet")
    B, S, H, W = 2, 2, 48, 160
    w = _window(B, S, H, W)
    twin = standins.PoseNetTwin(standins.posenet_params(int(g["seed"]))).cuda().eval()
    calls = []
    twin.register_forward_hook(lambda *a: calls.append(1))
    depths = [_t(w["depth_t"])] + [_t(w["depth_s"][i]) for i in range(S)]
    poses, poses_inv, out = train_mono.solve_pose_iteratively(4, depths, twin, _t(w["target"]), [_t(w["sources"][i]) for i in range(S)], _t(w["K"]),
                                                              return_errors=True)
    assert not calls                                                 # the torch module was not evaluated
    got = np.concatenate([torch.cat(poses).cpu().numpy(), torch.cat(poses_inv).cpu().numpy()])
    assert np.max(np.abs(got - g["loop_poses"])) < 2e-5 * np.abs(g["loop_poses"]).max()
    assert set(out["fwd"]) == {"diff_img", "img_rec", "valid_mask", "weight_mask", "poses", "auto_mask_error", "auto_mask"}
    assert np.max(np.abs(out["fwd"]["poses"].cpu().numpy() - g["loop_stacked"][:S * B])) < 2e-5 * np.abs(g["loop_stacked"]).max()


def test_posenet_from_a_reference_style_checkpoint(tmp_path):
    """PoseNetHIP.load_checkpoint: the 'pose_state_dict' of a checkpoint written the way utils/learning_helpers.py:20-27 writes it
    gives the golden poses of the reference's module with those parameters"""
    import standins
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine
    from tightly_coupled_sfm_amd.posenet import PoseNetHIP
    g = load_golden("posenet")
    H, W, N = 48, 160, 4
    (tmp_path / "best_model").mkdir()
    torch.save({"pose_state_dict": standins.PoseNetTwin(standins.posenet_params(int(g["seed"]))).state_dict(), "depth_state_dict": {},
                "best_val_loss": 0.0, "epoch": 1}, tmp_path / "best_model" / "best_model.pt")
    e = Engine(H, W, N)
    net = PoseNetHIP(e, N)
    net.load_checkpoint(str(tmp_path))
    b = synth.make_batch(N, H, W, seed0=40, both_directions=True)
    pose = net(_t(np.concatenate([b["tgt"], b["src"]], 1))).cpu().numpy()
    assert np.max(np.abs(pose - g["a_pose"])) < 1e-5 * np.abs(g["a_pose"]).max()


@pytest.mark.parametrize("S,lanes,wpc", [(1, 1, 1), (1, 3, 1), (2, 2, 1), (1, 2, 2), (1, 2, 5)])
def test_odometry_sequence_matches_per_window_calls(S, lanes, wpc):
    """tcsfm_odometry_sequence (per window: coupled PoseNet loop -> refinement, windows on the lanes, frames streamed once): the
    PoseNet poses and the refined poses equal, bit for bit, one solve_pose_iteratively + one refine_window call per window"""
    import standins
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd.posenet import PoseNetHIP
    H, W, T, IT = 48, 160, 12, 3
    seq = synth.make_sequence(T, H, W, seed=6)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    WB = max(1, wpc)
    e = Engine(H, W, 2 * S * WB, lanes=lanes)
    net = PoseNetHIP(e, 2 * S * WB, standins.posenet_params(5))
    o = default_opts(n_iters=3, argmin=1)
    ref_init, ref_out = [], []
    for c0 in range(0, T - S, WB):                        # the same batches, call by call
        nb = min(WB, T - S - c0)
        K = t(np.repeat(seq["K"][None], nb, 0)).cuda()
        tg, dt_ = t(seq["frames"][c0:c0 + nb]).cuda(), t(seq["depths"][c0:c0 + nb]).cuda()
        sr = torch.stack([t(seq["frames"][c0 + 1 + s:c0 + 1 + s + nb]) for s in range(S)]).cuda()
        ds_ = torch.stack([t(seq["depths"][c0 + 1 + s:c0 + 1 + s + nb]) for s in range(S)]).cuda()
        p0, _ = net.solve_pose_iteratively(IT, tg, sr, dt_, ds_, K)
        pr = e.refine_window(tg, sr, dt_, ds_, K, p0, o)[0]
        # stacked order of a call [fwd (s, b) | inv (s, b)] -> per window [fwd s | inv s]
        idx = [[s * nb + b for s in range(S)] + [S * nb + s * nb + b for s in range(S)] for b in range(nb)]
        ref_init += [p0[i].cpu() for i in idx]; ref_out += [pr[i].cpu() for i in idx]
    init, out = net.odometry_sequence(t(seq["frames"]).pin_memory(), t(seq["depths"]).pin_memory(), seq["K"], o, sources=S, iterations=IT,
                                      windows_per_call=wpc)
    assert torch.equal(init, torch.stack(ref_init)) and torch.equal(out, torch.stack(ref_out))
    assert not torch.equal(init, out)
    if WB > 1:      # against one window per call: the PoseNet's work split depends on the number of images (rounding only)
        i1, o1 = net.odometry_sequence(t(seq["frames"]).pin_memory(), t(seq["depths"]).pin_memory(), seq["K"], o, sources=S, iterations=IT,
                                       windows_per_call=1)
        assert float((i1 - init).abs().max()) < 1e-5 * float(init.abs().max())


def test_odometry_sequence_with_centred_windows_of_three_frames():
    """the reference's KITTI windows (3 frames, target in the middle, sources = previous and next frame) through
    tcsfm_odometry_sequence, two windows per call: equal to solve_pose_iteratively + refine_window on the hand-gathered batches"""
    import standins
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd.posenet import PoseNetHIP
    H, W, T, IT, S, WB = 48, 160, 11, 3, 2, 2
    seq = synth.make_sequence(T, H, W, seed=8)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    e = Engine(H, W, 2 * S * WB, lanes=2)
    net = PoseNetHIP(e, 2 * S * WB, standins.posenet_params(5))
    o = default_opts(n_iters=3, argmin=1)
    ref_init, ref_out = [], []
    for c0 in range(0, T - S, WB):
        nb = min(WB, T - S - c0)
        K = t(np.repeat(seq["K"][None], nb, 0)).cuda()
        tg, dt_ = t(seq["frames"][c0 + 1:c0 + 1 + nb]).cuda(), t(seq["depths"][c0 + 1:c0 + 1 + nb]).cuda()
        sr = torch.stack([t(seq["frames"][c0 + p:c0 + p + nb]) for p in (0, 2)]).cuda()
        ds_ = torch.stack([t(seq["depths"][c0 + p:c0 + p + nb]) for p in (0, 2)]).cuda()
        p0, _ = net.solve_pose_iteratively(IT, tg, sr, dt_, ds_, K)
        pr = e.refine_window(tg, sr, dt_, ds_, K, p0, o)[0]
        idx = [[s * nb + b for s in range(S)] + [S * nb + s * nb + b for s in range(S)] for b in range(nb)]
        ref_init += [p0[i].cpu() for i in idx]; ref_out += [pr[i].cpu() for i in idx]
    init, out = net.odometry_sequence(t(seq["frames"]).pin_memory(), t(seq["depths"]).pin_memory(), seq["K"], o, sources=S, iterations=IT,
                                      windows_per_call=WB, target_pos=-1)
    assert torch.equal(init, torch.stack(ref_init)) and torch.equal(out, torch.stack(ref_out))
