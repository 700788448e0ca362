#!/bin/bash
# GPU box: where the wave cycles of the row-owning FFN go (rocprofv3 --pmc, two SQ passes), old kernel beside it.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rowspmc
rm -rf $O; mkdir -p $O
for v in 1 0; do
FFN_ROWS=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/a$v -- python3 $R/tools/bench_ffn_only.py 512 0 > $O/a$v.log 2>&1
FFN_ROWS=$v rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b$v -- python3 $R/tools/bench_ffn_only.py 512 0 > $O/b$v.log 2>&1
FFN_ROWS=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $O/c$v -- python3 $R/tools/bench_ffn_only.py 512 0 > $O/c$v.log 2>&1
for d in a$v b$v c$v; do python3 $R/tools/pmc_summary.py $O/$d ffn; done
done > $O/summary.txt 2>&1
cat $O/summary.txt
