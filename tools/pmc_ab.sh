#!/bin/bash
# usage (GPU box): tools/pmc_ab.sh <tag> "<kernel substr>" <bench args...>  -> two PMC passes, per-kernel means
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; k=$2; shift; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${tag}A -- python3 $R/bench.py --no-extras "$@" > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}A $k
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${tag}B -- python3 $R/bench.py --no-extras "$@" > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}B $k
