#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "split_pair_decomposition or fp32_matches_reference or benched" > $O/gputest_subset.log 2>&1; tail -5 $O/gputest_subset.log
python tools/time_pos_ops.py split > $O/pos_ops.log 2>&1; grep "==" $O/pos_ops.log; grep -i "dual\|tail\|GEMM  " $O/pos_ops.log | head -12
for rep in 1 2; do
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/bench_300_$rep.json 2> $O/bench_300_$rep.err; cut -c80-130 $O/bench_300_$rep.json
SLIDE_GXS_CHAIN=0 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/bench_300_nochain_$rep.json 2> $O/bench_300_nochain_$rep.err; cut -c80-130 $O/bench_300_nochain_$rep.json
done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/bench_20.json 2> $O/bench_20.err; cut -c80-130 $O/bench_20.json
