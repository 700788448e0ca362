#!/bin/bash
# GPU box: HBM traffic of each workload's kernels from separate rocprofv3 --pmc passes -> gpurun_out/traffic/r04_traffic.json
# usage: tools/collect_traffic.sh <key e.g. ecg:512> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
key=$1; shift
tag=$(echo $key | tr ':' '_')
O=$R/gpurun_out/traffic
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_f -- python3 $R/bench.py --no-extras "$@" > $O/${tag}_f.log 2>&1 || { tail -5 $O/${tag}_f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_w -- python3 $R/bench.py --no-extras "$@" > $O/${tag}_w.log 2>&1 || { tail -5 $O/${tag}_w.log; exit 1; }
python3 $R/tools/traffic_json.py $key $O/${tag}_f $O/${tag}_w $O/r04_traffic.json
rm -rf $O/${tag}_f $O/${tag}_w
