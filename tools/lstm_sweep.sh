#!/bin/bash
# GPU box: config-4 (NASA LSTM) samples/s over the batch size -> gpurun_out/<tag>/B<batch>.json
R=$GRAFT_REPO_ROOT
tag=${1:-lstm_sweep}; shift
mkdir -p $R/gpurun_out/$tag
for B in "$@"; do
  steps=$(( B >= 4096 ? 10 : 40 ))
  timeout -k 10 300 python3 $R/bench.py --workload nasa_lstm --batch $B --steps $steps --warmup 2 --no-cpu-baseline > $R/gpurun_out/$tag/B$B.json 2> $R/gpurun_out/$tag/B$B.err || { tail -5 $R/gpurun_out/$tag/B$B.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/$tag/B$B.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("B=$B  %.1f samples/s  %.2f ms/step  | %s %.3f ms frac %.3f" % (d["value"], d["ms_per_step"], r.get("kernel"), r.get("ms_per_launch", 0), r.get("frac", 0)))
PY
done
