#!/bin/bash
# GPU box: where the wave cycles of the fused in-projection + attention kernel go (rocprofv3 --pmc, separate SQ passes of
# tools/attn_phases.py with NO_STAMPS=1: warm-up + 200 launches of the production kernel, nothing else on the device).
#   tools/collect_attn_pmc.sh [tag]      -> gpurun_out/attnpmc_<tag>/summary.txt   (FFD_TUNE is passed through)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-base}
O=$R/gpurun_out/attnpmc_$TAG
rm -rf $O; mkdir -p $O
export NO_STAMPS=1
for W in "ecg 512" "syn512 2048"; do
  set -- $W
  T=$1_$2
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/a_$T -- python3 $R/tools/attn_phases.py $1 $2 > $O/a_$T.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b_$T -- python3 $R/tools/attn_phases.py $1 $2 > $O/b_$T.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $O/c_$T -- python3 $R/tools/attn_phases.py $1 $2 > $O/c_$T.log 2>&1 || exit 1
  for d in a_$T b_$T c_$T; do echo "== $d"; python3 $R/tools/pmc_summary.py $O/$d qkv_attention; done
done > $O/summary.txt 2>&1
cat $O/summary.txt
