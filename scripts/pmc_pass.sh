#!/bin/bash
# One rocprofv3 counter pass over a short bench run (GPU box).  Usage:
#   bash scripts/pmc_pass.sh <tag> "<counters>" [extra bench args]
# Output: gpurun_out/pmc_<tag>/.../*_counter_collection.csv
set -e
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 1 --warmup 1 --cpu-reps 0 --mp-steps 3 "$@" > gpurun_out/pmc_$tag.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_$tag
