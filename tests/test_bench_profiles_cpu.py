"""bench.py's roofline block cites committed profile files by EXACT name (one tag): whenever those files exist they must be the
ones of the bench workload -- VERDICT r02: the line showed 19.3 MB of traffic because sorted(glob)[-1] picked another mode's file."""
import csv
import json
import os

from conftest import REPO


def test_cited_profiles_belong_to_the_bench_workload():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    alg = 32 * bench.H * bench.W * 2                     # config 2: fwd + inv directed pair per launch
    t = bench.load_pmc()
    if t is not None:
        assert alg <= t <= 1.5 * alg, (t, alg)            # 36 B/px actually read + halo re-reads: ~1.24x
        assert json.load(open(os.path.join(REPO, "profiles", f"{bench.PROFILE_TAG}_pmc_traffic.json")))["waves_per_linearize_launch"] == 2 * bench.H * bench.W / 64
    rp = bench.rocprof_avg_us()
    if rp:
        for key, v in rp.items():
            assert v["file"].startswith(f"profiles/{bench.PROFILE_TAG}_") and os.path.exists(os.path.join(REPO, v["file"]))
            rows = [r for r in csv.DictReader(open(os.path.join(REPO, v["file"]))) if bench.KERNEL in r["Name"]]
            assert len(rows) == 1 and abs(float(rows[0]["AverageNs"]) * 1e-3 - v["us"]) < 1e-3
        if "lanes_1" in rp:                                # one call in flight, the kernel has the chip: a B=1 launch takes 5 .. 20 us
            assert 5.0 < rp["lanes_1"]["us"] < 20.0
        if "saturated" in rp:                              # 64 directed pairs per launch
            assert 100.0 < rp["saturated"]["us"] < 400.0
