#!/bin/bash
# Run on the GPU box: the round's bench lines, rocprofv3 kernel summaries and PMC traffic passes -> gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
prof() {  # prof <name> <bench args...>: rocprofv3 --kernel-trace --stats of one bench run -> <name>_kernel_stats.csv
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py --no-extras "$@" > $O/${name}_profiled.json 2>/dev/null
  cp $O/prof_$name/*/*kernel_stats.csv $O/${name}_kernel_stats.csv && rm -rf $O/prof_$name
}
rm -f $R/gpurun_out/traffic/r04_traffic.json
python3 $R/bench.py --ablation > $O/bench_ecg_B512.json 2> $O/bench_ecg_B512.err
echo "ecg done: $(python3 -c "import json; d=json.loads(open('$O/bench_ecg_B512.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'])")"
prof ecg_B512
python3 $R/bench.py --cache --no-cpu-baseline > $O/bench_ecg_B512_cache.json 2>/dev/null
python3 $R/bench.py --workload syn512 --batch 8192 --cache --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_syn512_B8192_cache.json 2>/dev/null
python3 $R/bench.py --workload syn512 --batch 2048 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_syn512_B2048.json 2>/dev/null
prof syn512_B2048 --workload syn512 --batch 2048 --steps 3 --warmup 1
echo "syn512 done"
python3 $R/bench.py --workload nasa_lstm --batch 512 --steps 200 --warmup 10 > $O/bench_nasa_lstm_B512.json 2>/dev/null
python3 $R/bench.py --workload nasa_lstm --batch 8192 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_nasa_lstm_B8192.json 2>/dev/null
prof nasa_lstm_B512 --workload nasa_lstm --batch 512 --steps 20 --warmup 2
prof nasa_lstm_B8192 --workload nasa_lstm --batch 8192 --steps 5 --warmup 1
echo "lstm done"
bash $R/tools/collect_traffic.sh ecg:512 --steps 20 --warmup 2 > $O/traffic_ecg.log 2>&1
bash $R/tools/collect_traffic.sh syn512:2048 --workload syn512 --batch 2048 --steps 2 --warmup 1 > $O/traffic_syn.log 2>&1
bash $R/tools/collect_traffic.sh nasa_lstm:8192 --workload nasa_lstm --batch 8192 --steps 2 --warmup 1 > $O/traffic_lstm8192.log 2>&1
bash $R/tools/collect_traffic.sh nasa_lstm:512 --workload nasa_lstm --batch 512 --steps 5 --warmup 1 > $O/traffic_lstm512.log 2>&1
cp $R/gpurun_out/traffic/r04_traffic.json $O/r04_traffic.json
python3 $R/tools/batch_sweep.py 2>/dev/null | grep '^{"B"' > $O/batch_sweep.txt
python3 $R/tools/ffn_d_sweep.py 2>/dev/null | grep '^{' > $O/ffn_d_sweep.txt
python3 $R/tools/batch_sweep.py 8 10 12 16 20 24 28 32 40 44 50 56 64 66 80 100 128 160 200 2>/dev/null | grep '^{"B"' > $O/batch_sweep_small.txt
python3 $R/tools/sweep_tile_heights.py 28,32,40,44,50,56,64,66 2>/dev/null | grep '^B=' > $O/tile_height_sweep.txt
python3 $R/tools/lstm_trace.py 512 2>/dev/null | tail -1 > $O/lstm_trace_B512.json
python3 $R/tools/lstm_trace.py 2048 2>/dev/null | tail -1 > $O/lstm_trace_B2048.json
python3 $R/tools/attn_phases.py ecg 512 2>/dev/null > $O/attn_phases_ecg512.json
python3 $R/tools/attn_phases.py syn512 2048 2>/dev/null > $O/attn_phases_syn2048.json
python3 $R/tools/attn_phases.py ecg 512 0 2>/dev/null > $O/attn_phases_ecg512_pure.json
python3 $R/bench.py --gpus 2 --rehearse-one-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_rehearse_2ranks_one_gpu.json 2> $O/bench_rehearse.err
ls -la $O
