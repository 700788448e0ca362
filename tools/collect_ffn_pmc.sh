#!/bin/bash
# GPU box: PMC evidence for the dominant kernel (k_ffn_ln) -- pipe utilisation pass + separate FETCH / WRITE passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ffnpmc
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/bench_ffn_only.py 512 4 > $O/a.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/bench_ffn_only.py 512 4 > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/bench_ffn_only.py 512 4 > $O/w.log 2>&1
for d in a f w; do python3 $R/tools/pmc_summary.py $O/$d ffn_ln; done
python3 - <<PY
import csv, glob
f = glob.glob("$O/a/**/*kernel_trace.csv", recursive=True)[0]
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "ffn_ln" in r["Kernel_Name"]]
print("kernel duration under the profiler (us): mean %.1f min %.1f n=%d" % (sum(d)/len(d), min(d), len(d)))
PY
