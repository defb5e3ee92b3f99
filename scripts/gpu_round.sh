#!/bin/bash
# Full GPU check on the box: parity tests (the three matrix modes are covered in-process by the matmul_mode fixture), the bench line and a
# kernel-trace profile of the same bench command.   bash scripts/gpu_round.sh [tag]
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/tests_$tag.log 2>&1 || { tail -40 gpurun_out/tests_$tag.log; exit 1; }
tail -3 gpurun_out/tests_$tag.log
timeout -k 10 400 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -20 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-reps 0 --train-steps 0 --strict-steps 0 > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_$tag.csv
head -12 gpurun_out/kernel_stats_$tag.csv
