#!/bin/bash
# usage (on the GPU box): tools/prof_ab.sh <tag> <bench args...>   -> rocprofv3 kernel-trace stats of one bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --no-extras "$@" > $R/gpurun_out/$tag.json 2>/dev/null
python3 $R/tools/kstats.py $R/gpurun_out/$tag 6
