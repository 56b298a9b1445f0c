#!/bin/bash
# round-5 check of the split pair-decomposition plan (GPU box): parity tests, per-op times of the position plan variants, bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b; O=gpurun_out/r05b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "fp32_matches_reference or benched_arithmetic or benched_launch or full_chain_matches or sampler_matches or sampler_tail" > $O/gputest_subset.log 2>&1
tail -15 $O/gputest_subset.log
python tools/time_pos_ops.py split > $O/pos_ops_gxs.log 2>&1
SLIDE_PP=0 python tools/time_pos_ops.py split > $O/pos_ops_gxs_nopp.log 2>&1
grep "==" $O/pos_ops_*.log
cat $O/pos_ops_gxs.log
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity > $O/bench_300.json 2> $O/bench_300.err
for m in 1 3 4; do SLIDE_POS_MULT=$m python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/bench_300_mult$m.json 2> $O/bench_300_mult$m.err; done
SLIDE_PP=0 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/bench_300_nopp.json 2> $O/bench_300_nopp.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
for f in bench_300 bench_300_mult1 bench_300_mult3 bench_300_mult4 bench_300_nopp bench_steps20; do echo $f; cut -c1-200 $O/$f.json; tail -2 $O/$f.err; done
