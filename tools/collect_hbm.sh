#!/bin/bash
# GPU box: rocprofv3 kernel stats of the HBM-bound kernels at the config-5 shard size -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-hbm}
O=$R/gpurun_out/$tag
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/hbm_kernels.py ${2:-8192} ${3:-10} > $O/bytes.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python3 $R/tools/hbm_table.py $O/kernel_stats.csv $O/bytes.json | tee $O/hbm_table.csv
