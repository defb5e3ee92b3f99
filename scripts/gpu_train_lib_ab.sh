#!/bin/bash
# Same-box comparison of two builds of the library on the 250k-node training step: the in-tree build against the
# one named by $1 (default scripts/micro/base_libmgn_hip.so), two runs each, alternating.
base=${1:-scripts/micro/base_libmgn_hip.so}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/train_lib_ab.txt
for rep in 1 2; do
  for which in base tree; do
    if [ $which = base ]; then export MGN_LIB_PATH=$PWD/$base; else unset MGN_LIB_PATH; fi
    timeout -k 10 200 python scripts/bench_train.py --cpu-grid 0 --steps 5 --warmup 2 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    python - "$which" <<'PY' >> gpurun_out/train_lib_ab.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], "step %.2f ms  forward %.2f  backward %.2f  optimizer %.2f" % (d["ms_per_step"], d["ms_forward"], d["ms_backward"], d["ms_optimizer"]))
PY
  done
done
cat gpurun_out/train_lib_ab.txt
