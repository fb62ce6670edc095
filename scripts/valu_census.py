#!/usr/bin/env python3
"""VALU census of k_linearize: dynamic instruction mix of one wave from the compiler's ISA, priced per instruction class with the
MEASURED issue costs of scripts/valu_rate.hip (profiles/r02_valu_rate.jsonl), against the PMC instruction count and the measured
kernel durations.

    python scripts/valu_census.py [--kernel SUBSTR] [--out profiles/r02_valu_census.json]

Method.  hipcc -save-temps gives the kernel's ISA.  Control flow of this kernel is simple and is recovered structurally:
  * a backward branch closes a LOOP; its body executes `trips` times per wave.  The rolled neighbour loops of phase 2 have
    compile-time trip counts (pass A: 8 after the peeled first neighbour, pass B: 9; the window-selection variant has one more
    9-trip loop per other source), taken from --trips in program order;
  * a forward s_cbranch_execz / s_cbranch_scc* skips a REGION for the waves (or workgroups) whose condition fails; the fraction
    of waves that execute each region is given by --frac in program order (default: the phase-1 split of a 32x16 tile --
    2 of 8 waves run the two-pixel (tile + halo ring) sequence, 6 of 8 the one-pixel sequence; everything else 1 or 0).
Instruction classes and their cost in clocks per wave-instruction per SIMD come from the microbenchmark at 4 waves per SIMD
(the kernel's occupancy).  Unmeasured opcodes are priced at the half-rate cost and listed.
"""
import argparse, collections, json, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.build import FLAGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def isa_of(kernel_substr, line_tables=False):
    tmp = tempfile.mkdtemp(prefix="census_")
    src = os.path.join(ROOT, "tightly_coupled_sfm_amd", "csrc", "tcsfm_api.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + (["-gline-tables-only"] if line_tables else []) + [src, "-o",
                           os.path.join(tmp, "lib.so"), "-save-temps=obj"], cwd=tmp, stderr=subprocess.DEVNULL)
    s = open(os.path.join(tmp, "tcsfm_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    if line_tables:
        global FILES
        FILES = {int(m.group(1)): m.group(2) for l in s for m in [re.match(r'\s+\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]*)"', l)] if m}
    start = next(i for i, l in enumerate(s) if re.match(r"^_Z\w*" + re.escape(kernel_substr) + r"\w*:", l))
    end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
    return s[start:end + 1]


def load_costs(path):
    """class -> clocks per wave-instruction per SIMD at 4 waves / SIMD"""
    c = {}
    for l in open(path):
        r = json.loads(l)
        if r["waves_per_simd"] == 4:
            c[r["class"]] = r["clk_per_wave_inst_per_simd"]
    return c


def price_table(c):
    full = sum(c[k] for k in ("v_fma_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_fmac_f32", "v_add_u32", "v_and_b32", "v_ashrrev_i32")) / 8
    half = sum(c[k] for k in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_max_f32", "v_min_f32", "v_min_i32", "v_cvt_f32_i32",
                              "v_floor_f32", "v_cmp_gt_f32 vcc", "v_cndmask_b32_e64 sgpr-pair", "v_mov_b32_dpp", "v_mul_lo_u32")) / 12
    quarter = c["v_rcp_f32"]
    table = {}
    for op in ("v_fma_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
               "v_and_b32", "v_or_b32", "v_xor_b32", "v_ashrrev_i32", "v_lshlrev_b32", "v_lshrrev_b32", "v_not_b32", "v_bfi_b32",
               "v_add_co_u32", "v_addc_co_u32", "v_and_or_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_lshl_or_b32", "v_add3_u32",
               "v_mul_u32_u24", "v_mul_i32_i24", "v_bfe_u32", "v_bfe_i32", "v_or3_b32"):
        table[op] = ("full", full)
    table["v_mov_b32"] = ("mov", c["v_mov_b32"])
    table["v_accvgpr_write_b32"] = table["v_accvgpr_read_b32"] = ("mov", c["v_mov_b32"])
    for op in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32", "v_max_f32", "v_min_f32", "v_med3_f32", "v_max3_f32", "v_min3_f32",
               "v_max_i32", "v_min_i32", "v_max_u32", "v_min_u32", "v_med3_i32", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_cvt_f32_u32", "v_cvt_u32_f32",
               "v_floor_f32", "v_fract_f32", "v_trunc_f32", "v_rndne_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u32_u24",
               "v_mad_i32_i24", "v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32", "v_readfirstlane_b32", "v_readlane_b32", "v_writelane_b32",
               "v_cvt_f64_f32", "v_cvt_f32_f64"):
        table[op] = ("half", half)
    for op in ("v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32",
               "v_permlane32_swap_b32", "v_permlane16_swap_b32"):
        table[op] = ("quarter", quarter)
    return table, dict(full=full, half=half, quarter=quarter, mov=c["v_mov_b32"], cndmask_vcc_after_cmp=2 * c["v_cmp+v_cndmask pair (2 insts)"] - c["v_cmp_gt_f32 vcc"],
                       cndmask_vcc_back_to_back=c["v_cndmask_b32 vcc"])


def classify(op, operands, table, costs, prev_vcc_write):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return "half:v_cmp", costs["half"]
    if base == "v_cndmask_b32":
        if op.endswith("_e32") or operands.rstrip().endswith("vcc"):
            return "cndmask_vcc", costs["cndmask_vcc_after_cmp"]
        return "half:v_cndmask_e64", costs["half"]
    if op.endswith("_dpp"):
        return "half:dpp", costs["half"]
    if base in table:
        k, v = table[base]
        return f"{k}:{base}", v
    return f"UNMEASURED:{base}", costs["half"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_linearizeILi6ELb0ELi1ELi32ELi16ELi512ELb0ELb0ELb0E")
    ap.add_argument("--rates", default=os.path.join(ROOT, "profiles", "r02_valu_rate.jsonl"))
    ap.add_argument("--trips", default="1", help="trip counts of the loops in program order (rounds 2-3: 8,9 -- the rolled neighbour loops; the round-4 kernel has them unrolled)")
    ap.add_argument("--frac", default="", help="executing fraction of each skipped region in program order (see --list)")
    ap.add_argument("--list", action="store_true", help="print loops / regions with their line ranges and exit")
    ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "r01_g_pmc_summary.json"))
    ap.add_argument("--out", default="")
    ap.add_argument("--lines", type=int, default=0, help="also print the N source lines that cost most VALU clocks (compiles with -gline-tables-only)")
    a = ap.parse_args()
    isa = isa_of(a.kernel, a.lines > 0)
    labels = {l.split(":")[0]: i for i, l in enumerate(isa) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops, regions = [], []
    for i, l in enumerate(isa):
        m = re.match(r"\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if not m:
            continue
        tgt = labels[m.group(2)]
        if tgt <= i:
            loops.append((tgt, i))
        elif m.group(1) != "s_branch":
            regions.append((i + 1, tgt, m.group(1)))
    trips = [int(x) for x in a.trips.split(",")] if a.trips else []
    trips += [1] * (len(loops) - len(trips))       # loops not named (the group-reduction tail: its regions carry weight 0 in direct mode)
    fracs = [float(x) for x in a.frac.split(",")] if a.frac else None
    nvalu = lambda lo, hi: sum(1 for l in isa[lo:hi] if re.match(r"\s+v_", l))
    if a.list or fracs is None:
        print("loops  :", [(lo, hi, nvalu(lo, hi + 1)) for lo, hi in loops])
        for k, (lo, hi, kind) in enumerate(regions):
            print(f"region {k}: lines {lo}-{hi} ({kind}) VALU insts {nvalu(lo, hi)}")
        if a.list:
            return
        raise SystemExit("give --frac for the regions above")
    assert len(fracs) == len(regions), (len(fracs), len(regions))
    weight = [1.0] * len(isa)
    for (lo, hi, _), f in zip(regions, fracs):
        for i in range(lo, hi):
            weight[i] *= f
    for (lo, hi), t in zip(loops, trips):
        for i in range(lo, hi + 1):
            weight[i] *= t
    table, costs = price_table(load_costs(a.rates))
    dyn, clk = collections.Counter(), collections.Counter()
    other = collections.Counter()
    by_line, cur = collections.Counter(), None
    by_line_n = collections.Counter()
    for i, l in enumerate(isa):
        ml = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
        if ml:
            cur = (int(ml.group(1)), int(ml.group(2)))
            continue
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", l)
        if not m or weight[i] == 0:
            continue
        op, rest = m.group(1), m.group(2).split(";")[0]
        if op.startswith("v_"):
            cls, c = classify(op, rest, table, costs, False)
            dyn[cls] += weight[i]; clk[cls] += weight[i] * c
            by_line[cur] += weight[i] * c; by_line_n[cur] += weight[i]
        else:
            other[op.split("_")[0] + "_" + (op.split("_")[1] if "_" in op else "")] += weight[i]
    n_valu, busy = sum(dyn.values()), sum(clk.values())
    groups = collections.Counter(); gclk = collections.Counter()
    for k in dyn:
        groups[k.split(":")[0]] += dyn[k]; gclk[k.split(":")[0]] += clk[k]
    out = {"kernel": a.kernel, "valu_insts_per_wave_census": round(n_valu, 1), "predicted_valu_busy_clk_per_wave": round(busy, 1),
           "mean_clk_per_valu_inst": round(busy / n_valu, 3),
           "class_cost_clk": {k: round(v, 3) for k, v in costs.items()},
           "by_group": {k: {"insts": round(groups[k], 1), "clk": round(gclk[k], 1)} for k in sorted(groups, key=lambda k: -gclk[k])},
           "by_opcode": {k: {"insts": round(dyn[k], 1), "clk": round(clk[k], 1)} for k in sorted(dyn, key=lambda k: -clk[k])},
           "non_valu_insts_per_wave": {k: round(v, 1) for k, v in other.most_common(12)},
           "loops": [{"lines": [lo, hi], "trips": t, "valu_insts_in_body": nvalu(lo, hi + 1)} for (lo, hi), t in zip(loops, trips)],
           "regions": [{"lines": [lo, hi], "branch": kind, "fraction_of_waves": f, "valu_insts": nvalu(lo, hi)} for (lo, hi, kind), f in zip(regions, fracs)]}
    if os.path.exists(a.pmc):
        pm = json.load(open(a.pmc))
        k = next((v for name, v in pm.items() if "k_linearize<6, false, 1" in name and "SQ_INSTS_VALU" in v), None)
        if k:
            per_wave = k["SQ_INSTS_VALU"] / k["SQ_WAVES"]
            out["pmc"] = {"file": os.path.basename(a.pmc), "SQ_INSTS_VALU_per_wave": round(per_wave, 1), "census_over_pmc": round(n_valu / per_wave, 4),
                          "SQ_ACTIVE_INST_VALU_quadclk_per_wave": round(k["SQ_ACTIVE_INST_VALU"] / k["SQ_WAVES"], 1)}
    if a.lines:
        src_cache = {}
        for (f, ln), c in by_line.most_common(a.lines):
            name = FILES.get(f, "?")
            path = name if os.path.isabs(name) else os.path.join(ROOT, "tightly_coupled_sfm_amd", "csrc", os.path.basename(name))
            if path not in src_cache:
                src_cache[path] = open(path).read().splitlines() if os.path.exists(path) else []
            text = src_cache[path][ln - 1].strip()[:110] if 0 < ln <= len(src_cache[path]) else ""
            print(f"{c:8.1f} clk {by_line_n[(f, ln)]:7.1f} inst  {os.path.basename(name)}:{ln}  {text}")
        return
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
