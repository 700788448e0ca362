#!/bin/bash
# usage (on the GPU box): tools/collect_small.sh   -> round-2 small-batch evidence under gpurun_out/r2_small/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_small
mkdir -p $O
cd $R
python3 tools/bench_b1.py 2>/dev/null > $O/harness_regime_ms_per_step.txt
python3 tools/sweep_small.py 2>/dev/null > $O/threshold_sweep.txt
python3 tools/sweep_mid.py 2>/dev/null > $O/mid_batch_ffn_sweep.txt
bash tools/prof_small.sh r2_small/prof_b1 1 300 > $O/b1_kernels.txt
bash tools/prof_small.sh r2_small/prof_b1_cache 1 300 cache > $O/b1_cache_kernels.txt
bash tools/prof_small.sh r2_small/prof_b8 8 300 > $O/b8_kernels.txt
cp $O/prof_b1/*/*kernel_stats.csv $O/b1_kernel_stats.csv
cp $O/prof_b1_cache/*/*kernel_stats.csv $O/b1_cache_kernel_stats.csv
cd $R && python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
