#!/bin/bash
# Counter passes for the processor kernels at 1M nodes (GPU box).  One rocprofv3 --pmc run per
# pass (at most 8 SQ / 4 TCC / 2 GRBM counters each), summaries into gpurun_out/pmc_r02_<tag>.txt
#   bash scripts/pmc_round2.sh [tag-suffix]
set -e
sfx=${1:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
pass() {
    tag=$1; counters=$2
    rm -rf gpurun_out/pmc_$tag
    set +e
    timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 1 --warmup 1 --cpu-reps 0 --mp-steps 3 > gpurun_out/pmc_$tag.log 2>&1
    rc=$?
    set -e
    # a pass that was killed at its limit ends the script (nothing more on the GPU); an unknown counter name only skips the pass
    if [ $rc -ge 124 ]; then echo "pass $tag: killed (rc $rc)"; exit $rc; fi
    if [ $rc -ne 0 ]; then echo "pass $tag: rc $rc"; tail -3 gpurun_out/pmc_$tag.log; return 0; fi
    python scripts/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/pmc_r02_$tag$sfx.txt
    rm -rf gpurun_out/pmc_$tag
    grep -c mean gpurun_out/pmc_r02_$tag$sfx.txt
}
pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
pass sq2 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pass sq3 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM"
pass tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
