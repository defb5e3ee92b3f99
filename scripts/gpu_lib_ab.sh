#!/bin/bash
# Same-box comparison of two builds of the library: the in-tree build against the one named by $1 (default
# scripts/micro/base_libmgn_hip.so), alternating, two runs each of the headline rollout step (bench.py without its CPU,
# strict and training legs) and of the 250k-node training step.
base=${1:-scripts/micro/base_libmgn_hip.so}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/lib_ab.txt
for rep in 1 2; do
  for which in base tree; do
    if [ $which = base ]; then export MGN_LIB_PATH=$PWD/$base; else unset MGN_LIB_PATH; fi
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-reps 0 --train-steps 0 --strict-steps 0 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    timeout -k 10 200 python scripts/bench_train.py --cpu-grid 0 --steps 5 --warmup 2 > gpurun_out/ab_tmp2.json 2> gpurun_out/ab_tmp2.err || { tail -20 gpurun_out/ab_tmp2.err; exit 1; }
    python - "$which" <<'PY' >> gpurun_out/lib_ab.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
t = json.load(open("gpurun_out/ab_tmp2.json"))
k = d["kernel_ms"]
print(sys.argv[1], "rollout step %.2f ms  (edge kernel %.3f  node kernel %.3f  project %.3f)   training step %.2f ms  forward %.2f  backward %.2f"
      % (d["ms_per_step"], k["mp_edge"], k["mp_node"], k["mp_project"], t["ms_per_step"], t["ms_forward"], t["ms_backward"]))
PY
  done
done
cat gpurun_out/lib_ab.txt
