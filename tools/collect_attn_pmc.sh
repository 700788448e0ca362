#!/bin/bash
# GPU box: where the wave cycles of the fused attention kernel go (rocprofv3 --pmc, SQ passes) for attn_pv = $1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PV=${1:-0}
O=$R/gpurun_out/attnpmc$PV
rm -rf $O; mkdir -p $O
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --tune attn_pv=$PV"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/a -- $CMD > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b -- $CMD > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_WAVES SQ_LEVEL_WAVES --output-format csv -d $O/c -- $CMD > $O/c.log 2>&1
for d in a b c; do python3 $R/tools/pmc_summary.py $O/$d qkv_attention; done > $O/summary.txt 2>&1
cat $O/summary.txt
