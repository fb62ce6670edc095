#!/usr/bin/env python3
"""Compile the HIP library with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
demangled name, VGPRs, AGPRs, scratch bytes, spills, LDS bytes, occupancy.    python scripts/kernel_resources.py [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import build as B
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-Rpass-analysis=kernel-resource-usage"] + B.FLAGS + [B.SRC, "-o", "/tmp/_tcsfm_res.so"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
recs, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        recs.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in recs], capture_output=True, text=True).stdout.splitlines()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r, n in zip(recs, names):
    n = n.split("(")[0].replace("void tc::", "")
    if flt in n:
        print(f"{n:78s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} spill {r.get('VGPRs Spill', r.get('VGPR Spill','?')):>3s} LDS {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}")
