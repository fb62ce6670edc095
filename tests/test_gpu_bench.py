"""bench.py contract (one JSON line, the keys the driver reads) -- single process, and a 2-rank rehearsal of the
torch.distributed path on ONE card (gloo; RCCL refuses two ranks on the same device, the 8-GPU run is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline"}


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_single_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", "--cpu-sample", "1",
                        "--sat-windows", "4", "--modes-budget", "3", "--shim-sample", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 10 and d["value"] > 100
    assert d["unit"] == "frame-pairs/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.01 * d["value"]                       # one window per step
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["launches"] == 4 * 300
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["algorithmic_bytes_per_launch"] == 32 * 192 * 640 * 2
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["avg_launch_us"] * 1e-3) < 0.01 * rf["achieved"]
    # the headline is THIS run's in-kernel bracket; the HIP event pair reads above it; the committed rocprof average of
    # the same kernel stands beside it and the two must agree (15 % after the ~1.2 us dispatch offset) -- frac_consistent
    assert "in-kernel" in rf["frac_source"] and rf["bound_actual"] == "valu"
    assert 3.0 < rf["avg_launch_us"] < rf["avg_launch_us_hip_events"] < rf["avg_launch_us"] + 8.0
    rc = rf["rocprof_committed"]
    if rc is not None:
        assert "lanes1_kernel_stats.csv" in rc["file"] and abs(rc["frac"] - rf["algorithmic_bytes_per_launch"] / rc["avg_launch_us"] * 1e-3 / 8000.0) < 1e-4
        assert rf["frac_consistent"] is True, (rf["avg_launch_us"], rc)
    if rf["valu_bound"] is not None:
        assert rf["valu_frac_of_bound"] == rf["valu_bound"]["frac_of_bound"] and 0.2 < rf["valu_frac_of_bound"] < 1.05
    # the steps rotate over a ring of distinct windows larger than the Infinity Cache; the hot-cache protocol is reported beside it
    assert d["config"]["ring_calls"] >= 64 and d["config"]["ring_input_MB"] > 268.0
    assert d["hot_cache"]["value"] > 100 and 0.8 < d["hot_cache"]["value_over_ring_value"] < 1.5
    if rf["traffic"] is not None:       # HBM bytes per launch of THIS workload (r02 cited another mode's file): at most 1.5x the algorithmic bytes, and at
        # least the two bordered image packs + the error plane a window's two pairs share since round 5 (every image packed once)
        floor = 2 * 194 * 642 * 16 + 192 * 640 * 4
        assert floor <= rf["traffic"] <= 1.5 * rf["algorithmic_bytes_per_launch"], rf["traffic"]
    assert d["config"]["lanes"] == 4 and d["single_stream"]["value"] > 100 and d["value"] > 0.9 * d["single_stream"]["value"]
    # the steps also ran as queued calls merged by the library (10 per launch sequence): same bits; the faster way is the headline
    mg, ln = d["launch_mode"]["merged"], d["launch_mode"]["lanes"]
    assert mg["calls_per_sequence"] == 10 and mg["same_poses"] is True and mg["value"] > 100 and ln["calls_in_flight"] == 4
    assert abs(d["value"] - max(mg["value"], ln["value"])) < 0.01 * d["value"] and d["config"]["steps_in_flight"] == (10 if mg["timed"] else 4)
    assert d["timed_blocks"] >= 1 and d["ms_per_step_blocks"]["min"] <= d["ms_per_step"] <= d["ms_per_step_blocks"]["max"]
    assert cb["one_thread"]["value"] > 0 and cb["one_thread"]["cores"] == 1
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert "workload" in d["config"] and d["roofline_saturated"]["achieved"] > rf["achieved"]
    assert d["roofline_saturated"]["frac_consistent"] in (True, None)
    # both launch modes ran (the faster one in the timed region, chosen on untimed probe blocks) and gave the same bits
    lm = d["launch_mode"]
    ot = lm["other_mode"]
    assert ("graph replay" in ln["mode"]) != ("graph replay" in ot["mode"]) and "probe" in lm["chosen_by"]
    assert ("merged sequences" in lm["timed"]) == mg["timed"]
    assert ot["same_poses"] is True and ot["captures"] >= d["config"]["ring_calls"] and ot["replays"] >= 60 - 6 and ot["value"] > 100
    assert d["host_enqueue_us_per_step"] > 0 and ot["host_enqueue_us_per_step"] > 0
    # the accuracy check on pairs rendered through the reference's own sampling model (the residual's minimiser is the scene truth there):
    # 4 GN iterations move the poses towards it, 16 LM iterations reach about 1 % / 0.02 degrees (tests/test_gpu_truth.py asserts the bar)
    # the driver's own line names the timed mode, brackets ITS launches, and carries a number for every BASELINE config,
    # for the reference-loss dense mode and for the Python shim
    assert "timed as" in d["config"]["workload"] and d["config"]["timed_as"] in d["config"]["workload"]
    tm = rf["timed_mode"]
    if mg["timed"]:
        assert "queued" in d["config"]["timed_as"] and tm["pairs_per_launch"] == 20 and 0 < tm["frac"] <= 1 and tm["launches"] >= 4
        assert abs(tm["frac"] - tm["algorithmic_bytes_per_launch"] / tm["avg_launch_us"] * 1e-3 / 8000.0) < 1e-4
        # the chip's rate on the kernel in that mode: all launches' bytes / the union of their in-kernel intervals (the two streams overlap)
        cl = tm["chip_level"]
        assert 1.0 <= cl["overlap"] <= 2.05 and tm["frac"] <= cl["frac"] <= 1 and cl["busy_us"] <= cl["launch_us_summed"] + 1e-6
    else:
        assert tm is None and "lanes" in d["config"]["timed_as"]
    md = d["modes"]
    want = {"config4_pose_scale_8it_640x192", "kitti_window_S2_pose_reference_rule_640x192", "kitti_window_S2_reference_loss_dense_640x192",
            "config5_dense_schur_320x240", "config5_dense_schur_448x256", "config5_reference_loss_full_320x240", "config5_reference_loss_quarter_320x240",
            "config5_reference_loss_full_448x256", "config5_reference_loss_quarter_448x256",
            "kitti_window_S2_reference_loss_dense_minibatch6_640x192"}      # (the reference driver's minibatch: the chip-filling figure of the mirror's default mode)
    assert want <= set(md)
    for k in want:
        m = md[k]
        assert m["us_per_call"] > 0 and m["frame_pairs_per_s"] > 100 and 0 < m["whole_call"]["frac"] <= 1 and 0 < m["dominant_kernel"]["frac"] <= 1, (k, m)
        assert m["whole_call"]["frac"] < m["dominant_kernel"]["frac"] * 1.5 + 0.05
    m6, m1 = md["kitti_window_S2_reference_loss_dense_minibatch6_640x192"], md["kitti_window_S2_reference_loss_dense_640x192"]
    assert m6["windows_per_call"] == 6 and m6["windows_per_s"] > 1.2 * m1["calls_per_s"]        # a minibatch fills the chip: well above one window per call
    sh = d["shim"]
    for k in ("pose", "pose_depth_reference_loss"):
        assert sh[k]["us_per_window"] > sh[k]["engine_call_us"] > 0 and sh[k]["windows"] >= 5
    tr = d["check"]["truth_sampler_consistent"]
    assert tr["gn_4_iterations"]["translation_rel"] < 0.8 * tr["initial"]["translation_rel"]
    assert tr["lm_16_iterations"]["translation_rel"] < 0.02 and tr["lm_16_iterations"]["rotation_deg"] < 0.05


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_one_card(tmp_path):
    """N > 1 runs BASELINE config 3's shard: 8 windows (16 directed pairs) per rank and step; the gathered poses equal what a
    single process computes for every rank's windows, bit for bit"""
    import numpy as np
    import torch
    env = dict(os.environ, TCSFM_BENCH_BACKEND="gloo", TCSFM_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    dump = str(tmp_path / "poses.npy")
    # started the way the driver starts it: `python bench.py --gpus 2` -- the parent launches the ranks itself
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--ring-mb", "100",
                        "--dump-poses", dump], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)                                                                       # rank 0 prints, once
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and d["value"] > 100
    assert d["config"]["collective_backend"] == "gloo" and d["config"]["collective_world_size"] == 2
    assert "in-kernel" in d["roofline"]["frac_source"] and d["roofline"]["frac"] > 0.01
    assert d["config"]["windows_per_gpu"] == 8 and d["config"]["directed_pairs_per_step"] == 16 and d["config"]["global_batch_frame_pairs"] == 16
    assert abs(d["value"] - 2 * 8 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]               # whole-job rate: 16 windows per step
    assert len(d["roofline"]["per_rank"]) == 2 and {x["rank"] for x in d["roofline"]["per_rank"]} == {0, 1}
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 32 * 192 * 640 * 16 and d["final_gather_us"] > 0
    got = np.load(dump)
    assert got.shape == (2, 16, 6)
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    e = Engine(192, 640, 16)
    for rank in range(2):          # the same windows in ONE process
        b = synth.make_batch(16, 192, 640, seed0=100 * rank, both_directions=True)
        t = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        pose, _, _ = e.refine(t["tgt"], t["src"], t["depth_t"], t["depth_s"], t["K"], t["pose_init"], default_opts(n_iters=4))
        pose = pose.cpu().numpy()           # pair form: (fwd, inv) interleaved; the bench's window form: all fwd pairs, then all inv pairs
        assert np.array_equal(np.concatenate([pose[0::2], pose[1::2]]), got[rank])


def test_bench_strong_scaling_option():
    """--total-windows fixes the job's windows per step (split over the ranks) and reports "scaling": "strong" (SURVEY 8e asks for both)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--cpu-sample", "0", "--sat-windows", "0",
                        "--total-windows", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["windows_per_gpu"] == 4 and d["config"]["directed_pairs_per_step"] == 8
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]


def test_bench_world_size_one_over_rccl():
    """the RCCL code path of bench.py on hardware: one rank, backend nccl (= RCCL) -- init_process_group, the all_reduce of the
    block count / block times, the final all_gather of the poses and the barriers all run on the card (the 8-GPU run itself is
    the driver's: only one-GPU boxes exist here)"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TCSFM_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-sample", "0",
                        "--sat-windows", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 100 and d["final_gather_us"] is not None and d["final_gather_us"] > 0
    assert d["config"]["collective_backend"] == "nccl" and d["config"]["collective_world_size"] == 1


def test_sequence_sharded_two_ranks_one_card(tmp_path):
    """the window loop of ONE sequence over two ranks (gloo rehearsal on one card) = the single-process sequence, bit for bit;
    and parallel.refine_sharded wired to the real Engine.refine"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    env = dict(os.environ, TCSFM_BENCH_BACKEND="gloo", TCSFM_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    dump = str(tmp_path / "seq.npy")
    common = [os.path.join(ROOT, "examples", "run_sequence_sharded.py"), "--frames", "37", "--sources", "2", "--height", "96", "--width", "320",
              "--windows-per-call", "4", "--dump", dump]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port())] + common, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(dump)
    import run_sequence_sharded as RS
    from tightly_coupled_sfm_amd import parallel
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    frames, depths, K, init = RS.make_sequence(37, 96, 320, 2)
    e = Engine(96, 320, 16, lanes=2)
    one = e.refine_sequence(frames, depths, K, init, default_opts(n_iters=4), sources=2, windows_per_call=4)
    assert got.shape == (35, 4, 6) and np.array_equal(got, one.numpy())
    # the same with the PoseNet loop inside (tcsfm_odometry_sequence per rank): equal to the single-process run, bit for bit
    dump2 = str(tmp_path / "odo.npy")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port())] + common[:-1] + [dump2, "--odometry", "2"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import standins
    from tightly_coupled_sfm_amd.posenet import PoseNetHIP
    net = PoseNetHIP(e, 16, standins.posenet_params(0))
    _, one_odo = net.odometry_sequence(frames, depths, K, default_opts(n_iters=4), sources=2, iterations=2, windows_per_call=4)
    assert np.array_equal(np.load(dump2), one_odo.numpy())
    # the DENSE mode of the sequence over two ranks -- poses gathered, every rank keeps (and writes) the depth maps of its
    # block; and gathered onto rank 0: poses and every pixel of every map equal to the single-process tcsfm_refine_dense_sequence
    import glob
    od = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
    one_p, one_d = e.refine_dense_sequence(frames, depths, K, init, od, sources=2, windows_per_call=4)
    for extra in ([], ["--gather-depths"]):
        dump3, dd = str(tmp_path / f"dense{len(extra)}.npy"), str(tmp_path / f"maps{len(extra)}")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port())] + common[:-1] + [dump3, "--dense", "--dump-depths", dd] + extra, capture_output=True, text=True,
                           timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert np.array_equal(np.load(dump3), one_p.numpy())
        files = sorted(glob.glob(dd + ".*.npy"), key=lambda f: int(os.path.basename(f).split(".")[1].split("-")[0]))
        assert len(files) == (1 if extra else 2)
        maps = np.concatenate([np.load(f) for f in files])
        assert maps.shape == tuple(one_d.shape) and np.array_equal(maps, one_d.numpy())
    # the pair-form helper on the real engine (single process: the gather is the identity)
    from tightly_coupled_sfm_amd import synth
    b = synth.make_batch(6, 96, 320, seed0=5, both_directions=True)
    t = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    e2 = Engine(96, 320, 6)
    fn = lambda tgt, src, depth_t, depth_s, K, pose: e2.refine(tgt, src, depth_t, depth_s, K, pose, default_opts(n_iters=2))[0]
    out = parallel.refine_sharded(fn, dict(tgt=t["tgt"], src=t["src"], depth_t=t["depth_t"], depth_s=t["depth_s"], K=t["K"], pose=t["pose_init"]), 6)
    assert torch.equal(out, fn(t["tgt"], t["src"], t["depth_t"], t["depth_s"], t["K"], t["pose_init"]))
