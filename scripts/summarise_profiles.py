#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (from scripts/collect_profiles.sh) into profiles/<tag>_*: kernel stats CSV, a PMC summary
and the HBM-traffic JSON bench.py reads (FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM prescribes for gfx950)."""
import collections, csv, glob as _glob, json, os, shutil, sys


class glob:          # newest first: a directory that still holds an earlier collection's files is summarised from the latest ones
    @staticmethod
    def glob(pat):
        return sorted(_glob.glob(pat), key=os.path.getmtime, reverse=True)


tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
ks1 = glob.glob(os.path.join(src, "trace_lanes1", "*", "*kernel_stats.csv"))
if ks1:
    shutil.copy(ks1[0], os.path.join(dst, f"{tag}_lanes1_kernel_stats.csv"))
kss = glob.glob(os.path.join(src, "trace_sat", "*", "*kernel_stats.csv"))
if kss:
    shutil.copy(kss[0], os.path.join(dst, f"{tag}_sat_kernel_stats.csv"))
for sub, name in (("_dref", "dref"), ("_dref_q", "dref_quarter")):       # dense mode on the reference's loss (scripts/collect_r04.sh)
    kd = glob.glob(os.path.join(root, "gpurun_out", tag + sub, "*", "*kernel_stats.csv"))
    if kd:
        shutil.copy(kd[0], os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
for name in ("dense_ref_timing.jsonl", "bench_modes.jsonl", "seq_cache_timing.jsonl"):
    if os.path.exists(os.path.join(root, "gpurun_out", f"{tag}_{name}")):
        shutil.copy(os.path.join(root, "gpurun_out", f"{tag}_{name}"), os.path.join(dst, f"{tag}_{name}"))
for name in ("bench.json", "bench_k20.json", "meta.json"):      # meta.json: hash of the kernel sources the profiles were collected on
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
pmc = collections.defaultdict(dict)
newest = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    newest.setdefault(os.path.dirname(f), f)
for f in newest.values():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if "tc::" in k:
            for c, x in v.items():
                pmc[k.split("(")[0]][c] = sum(x) / len(x)
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(pmc, f, indent=1, sort_keys=True)
lin = next((v for k, v in pmc.items() if "k_linearize" in k and "FETCH_SIZE" in v), None)
if lin:
    fetch_kb, write_kb = lin["FETCH_SIZE"], lin.get("WRITE_SIZE", 0.0)
    traffic = {"hbm_bytes_per_linearize_launch": int((2 * fetch_kb + write_kb) * 1024),
               "valu_insts_per_linearize_launch": lin.get("SQ_INSTS_VALU"), "waves_per_linearize_launch": lin.get("SQ_WAVES"),
               "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
               "note": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE; "
                       "separate rocprofv3 --pmc passes of `python bench.py --lanes 1 --coalesce 0 --graph-replay 0 --steps 20 --warmup 5`; per-launch average"}
    # the whole call: pack + 4 x (linearise + solve), every kernel's own per-launch average (same correction)
    hb = lambda v: (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024
    pack = next((v for k, v in pmc.items() if "k_pack" in k and "FETCH_SIZE" in v), None)
    solve = next((v for k, v in pmc.items() if "k_solve" in k and "FETCH_SIZE" in v), None)
    if pack and solve:
        traffic["hbm_bytes_per_call"] = int(hb(pack) + 4 * (hb(lin) + hb(solve)))
        traffic["hbm_bytes_per_pack_launch"] = int(hb(pack)); traffic["hbm_bytes_per_solve_launch"] = int(hb(solve))
    with open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(traffic)
# per-kernel HBM traffic (same correction) for every kernel that has the counters: the mode profiles have other dominant kernels
for k, v in pmc.items():
    if "FETCH_SIZE" in v:
        v["hbm_bytes_per_launch"] = int((2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024)
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(pmc, f, indent=1, sort_keys=True)
for k, v in pmc.items():
    print(k, {c: round(x, 1) for c, x in sorted(v.items())})
