#!/bin/bash
# Experiment (GPU box): the 1M-node mesh with its nodes in row-major, Morton and RCM order: step and kernel times, then
# FETCH_SIZE / WRITE_SIZE of the edge kernel per order (one rocprofv3 --pmc run per counter and order).
#   bash scripts/exp_reorder_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/reorder
for o in none morton rcm; do
    timeout -k 10 300 python bench.py --reorder $o --train-steps 0 --cpu-reps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$o', 'ms_per_step %.2f' % d['ms_per_step'], d['kernel_ms'])"
done
for o in none morton; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf gpurun_out/reorder/pmc
        timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/reorder/pmc -- python3 bench.py --reorder $o --steps 1 --warmup 1 --cpu-reps 0 --train-steps 0 --mp-steps 3 > gpurun_out/reorder/pmc.log 2>&1
        echo "order $o counter $c"; python scripts/pmc_summary.py gpurun_out/reorder/pmc | grep "mp_edge_h4\|mp_node_h" | grep -v duration
    done
done
