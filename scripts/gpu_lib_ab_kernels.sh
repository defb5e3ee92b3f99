#!/bin/bash
# Same-box, per-kernel comparison of two builds of the library: kernel-trace statistics of the 250k-node training step
# and of three headline rollout steps, for the build named by $1 (default scripts/micro/base_libmgn_hip.so) and for the
# in-tree build.  rocprofv3 runs python3 directly; the library is chosen through MGN_LIB_PATH in this shell.
base=${1:-scripts/micro/base_libmgn_hip.so}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/lib_ab_kernels.txt
for which in base tree base tree; do
  if [ $which = base ]; then export MGN_LIB_PATH=$PWD/$base; else unset MGN_LIB_PATH; fi
  for job in train rollout; do
    rm -rf gpurun_out/prof_ab
    if [ $job = train ]; then
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python3 scripts/bench_train.py --cpu-grid 0 --steps 3 --warmup 1 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    else
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 3 --warmup 1 --cpu-reps 0 --train-steps 0 --strict-steps 0 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    fi
    f=$(find gpurun_out/prof_ab -name "*kernel_stats.csv" | head -1)
    python3 - "$which" "$job" "$f" <<'PY' >> gpurun_out/lib_ab_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[3])))
keep = ("train_linear_bwd3_kernel<true, true>", "train_edge_bwd", "mp_edge_h4", "mp_node_h_kernel<true, true>", "train_linear_fwd_kernel<false, true>", "segment_sum", "encode_kernel<2", "encode_kernel<3", "decode_kernel", "mp_project")
out = []
for r in rows[:12]:
    for k in keep:
        if k in r["Name"]:
            out.append("%s %d x %.1f us" % (k.split("<")[0].replace("_kernel", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e3))
print(sys.argv[1], sys.argv[2], "|", "; ".join(out))
PY
  done
done
rm -rf gpurun_out/prof_ab
cat gpurun_out/lib_ab_kernels.txt
