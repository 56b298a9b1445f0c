cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "split_pair_decomposition or fp32_matches_reference or benched" > $O/gputest_subset.log 2>&1; tail -5 $O/gputest_subset.log
bash tools/rocprof_run5.sh r05a > $O/prof.log 2>&1; tail -6 $O/prof.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_steps20.json 2> $O/bench20.err; cut -c1-220 $O/r05_bench_steps20.json
python bench.py > $O/r05_bench_noflags.json 2> $O/bench.err; cut -c1-220 $O/r05_bench_noflags.json
