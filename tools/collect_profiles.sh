#!/bin/bash
# Run on the GPU box: refresh the round's bench lines and rocprofv3 kernel summaries under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
python3 $R/bench.py > $O/bench_ecg_B512.json 2> $O/bench_ecg_B512.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ecg -- python3 $R/bench.py --no-extras > $O/bench_ecg_B512_profiled.json 2>/dev/null
python3 $R/bench.py --cache --no-extras > $O/bench_ecg_B512_cache.json 2>/dev/null
python3 $R/bench.py --workload syn512 --batch 8192 --cache --steps 3 --warmup 1 --no-extras > $O/bench_syn512_B8192_cache.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_syn -- python3 $R/bench.py --workload syn512 --batch 2048 --steps 3 --warmup 1 --no-extras > $O/bench_syn512_B2048.json 2>/dev/null
python3 $R/bench.py --workload nasa_lstm --batch 512 --steps 200 --warmup 10 --no-extras > $O/bench_nasa_lstm_B512.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lstm -- python3 $R/bench.py --workload nasa_lstm --batch 512 --steps 20 --warmup 2 --no-extras > /dev/null 2>&1
for d in prof_ecg prof_syn prof_lstm; do cp $O/$d/*/*kernel_stats.csv $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
ls -la $O
