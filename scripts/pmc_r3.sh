#!/bin/bash
# Traffic and occupancy counter passes of the processor kernels at 1M nodes (GPU box): one
# rocprofv3 --pmc run per pass, per-kernel means into gpurun_out/pmc_r03_<pass>.txt (edge and node kernels both listed).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
pass() {
    tag=$1; counters=$2
    rm -rf gpurun_out/pmc_$tag
    set +e
    timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 1 --warmup 1 --cpu-reps 0 --mp-steps 3 --train-steps 0 --strict-steps 0 > gpurun_out/pmc_$tag.log 2>&1
    rc=$?
    set -e
    if [ $rc -ge 124 ]; then echo "pass $tag: killed (rc $rc)"; exit $rc; fi
    if [ $rc -ne 0 ]; then echo "pass $tag: rc $rc"; tail -3 gpurun_out/pmc_$tag.log; return 0; fi
    python scripts/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/pmc_r03_$tag.txt
    rm -rf gpurun_out/pmc_$tag
    grep -c mean gpurun_out/pmc_r03_$tag.txt
}
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
