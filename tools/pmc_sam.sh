#!/bin/bash
# rocprofv3 PMC passes over ONE eager SAM 2.1-L B=16 forward (tools/one_step.py): SQ counters per kernel -> gpurun_out/<tag>/pmc_*.txt
set -o pipefail
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]*(MFMA|LDS|WAIT|BUSY|WAVE|VALU)[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
run() { # name counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -o p -- python3 tools/one_step.py sam2l 1 > $O/$n.log 2>&1
  echo "$n rc=$?"
  python3 tools/pmc_sum.py $O/$n > $O/pmc_$n.txt 2>&1
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY
run b SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL
run c GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
grep -E "attn_|tok_linear|hiera_mlp|gemm256" $O/pmc_a.txt | head -60
