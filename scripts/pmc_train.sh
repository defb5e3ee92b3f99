#!/bin/bash
# Counter passes for the training kernels at 250k nodes (GPU box): one rocprofv3 --pmc run per pass over a short
# bench_train.py run, summaries into gpurun_out/pmc_train_<tag>.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
pass() {
    tag=$1; counters=$2
    rm -rf gpurun_out/pmc_tr_$tag
    set +e
    timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/pmc_tr_$tag -- python3 scripts/bench_train.py --cpu-grid 0 --steps 1 --warmup 1 --mp-steps 3 > gpurun_out/pmc_tr_$tag.log 2>&1
    rc=$?
    set -e
    if [ $rc -ge 124 ]; then echo "pass $tag: killed (rc $rc)"; exit $rc; fi
    if [ $rc -ne 0 ]; then echo "pass $tag: rc $rc"; tail -3 gpurun_out/pmc_tr_$tag.log; return 0; fi
    python3 scripts/pmc_summary.py gpurun_out/pmc_tr_$tag > gpurun_out/pmc_train_$tag.txt
    rm -rf gpurun_out/pmc_tr_$tag
    grep -c mean gpurun_out/pmc_train_$tag.txt
}
pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
pass lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES"
