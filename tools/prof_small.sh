#!/bin/bash
# usage (on the GPU box): tools/prof_small.sh <tag> <B> <steps> [cache]  -> per-kernel stats of one small-batch sampler
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/profile_small.py "$@" > $R/gpurun_out/$tag.log 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/$tag 12
