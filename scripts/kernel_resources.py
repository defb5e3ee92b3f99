#!/usr/bin/env python3
"""Register / LDS / spill / occupancy table of every kernel of libmgn_hip.so, from hipcc's
-Rpass-analysis=kernel-resource-usage (runs on the build box, no GPU needed).
    python scripts/kernel_resources.py > profiles/<round>_kernel_resources.txt"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gnn_amd", "csrc")
import sys  # noqa: E402
sys.path.insert(0, SRC)
from build import EXTRA_FLAGS  # noqa: E402  (the per-file flags of the product build)
rows = []
for f in ("mgn_kernels.hip", "mgn_train.hip", "mgn_graph.hip", "mgn_halo.hip", "mgn_coarse.hip"):
    res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] + EXTRA_FLAGS.get(f, []) +
                         ["-c", "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, os.path.join(SRC, f)],
                         cwd=SRC, capture_output=True, text=True)
    cur = None
    for line in res.stderr.splitlines():
        m = re.search(r"remark: (?:\s*)([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z ]*): (.*?) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            name = subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()
            cur = {"file": f, "kernel": name.replace("(anonymous namespace)::", "")}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
rows = [r for r in rows if "cub" not in r["kernel"] and "rocprim" not in r["kernel"]]
print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch B':>9s} {'LDS B':>7s} {'waves/SIMD':>10s}")
for r in rows:
    print(f"{r['kernel'][:78]:78s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>9s} "
          f"{r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>10s}")
