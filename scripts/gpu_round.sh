#!/bin/bash
# Full GPU check on the box: parity tests (the three matrix modes are covered in-process by the matmul_mode fixture), the bench line, a kernel-trace
# profile of the same bench command and the two HBM traffic counter passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tests_default.log 2>&1
tail -2 gpurun_out/tests_default.log
timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 3 --warmup 1 --cpu-reps 0 --train-steps 0 > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
bash scripts/pmc_pass.sh fetch "FETCH_SIZE" | grep mp_edge
bash scripts/pmc_pass.sh write "WRITE_SIZE" | grep mp_edge
