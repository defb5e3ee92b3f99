#!/bin/bash
# Training-path check on the box: training parity tests, the 250k-node training bench and a kernel-trace profile of it.
#   bash scripts/gpu_train_ab.sh [tag]
tag=${1:-ab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_training.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_train_$tag.log 2>&1 || { tail -30 gpurun_out/t_train_$tag.log; exit 1; }
tail -2 gpurun_out/t_train_$tag.log
timeout -k 10 200 python scripts/bench_train.py --cpu-grid 0 --steps 5 --warmup 2 > gpurun_out/bench_train_$tag.json 2> gpurun_out/bench_train_$tag.err || { tail -20 gpurun_out/bench_train_$tag.err; exit 1; }
cat gpurun_out/bench_train_$tag.json
rm -rf gpurun_out/prof_train_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$tag -- python3 scripts/bench_train.py --cpu-grid 0 --steps 3 --warmup 1 > gpurun_out/bench_train_prof_$tag.json 2> gpurun_out/bench_train_prof_$tag.err || exit 1
find gpurun_out/prof_train_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/train_kernel_stats_$tag.csv
find gpurun_out/prof_train_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/train_kernel_trace_$tag.csv
rm -rf gpurun_out/prof_train_$tag
head -6 gpurun_out/train_kernel_stats_$tag.csv | cut -c1-150
