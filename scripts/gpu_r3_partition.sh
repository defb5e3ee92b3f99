#!/bin/bash
# GPU box: partitioned-path tests (thread-ranks and 2-process gloo on one GPU) and the host-issue profile.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 560 python -u -m pytest tests/test_gpu_partition.py tests/test_gpu_dist.py -x -v -s --timeout 300 --timeout-method thread > gpurun_out/r3_partition_tests.log 2>&1 || { tail -40 gpurun_out/r3_partition_tests.log; exit 1; }
tail -15 gpurun_out/r3_partition_tests.log
timeout -k 10 200 python -u scripts/prof_host_partition.py --transport copy > gpurun_out/r3_prof_host_partition_copy.txt 2>&1 || { tail -30 gpurun_out/r3_prof_host_partition_copy.txt; exit 1; }
cat gpurun_out/r3_prof_host_partition_copy.txt
timeout -k 10 200 python -u scripts/prof_host_partition.py --transport nccl --profile > gpurun_out/r3_prof_host_partition_nccl.txt 2>&1 || { tail -30 gpurun_out/r3_prof_host_partition_nccl.txt; exit 1; }
head -60 gpurun_out/r3_prof_host_partition_nccl.txt
