cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_evidence; mkdir -p $O
bash tools/rocprof_modules.sh r06 > $O/profm.log 2>&1; tail -4 $O/profm.log
bash tools/decode_hbm.sh r06 > $O/dechbm.log 2>&1; tail -2 $O/dechbm.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_steps20.json 2> $O/bench20.err; cut -c1-200 $O/r06_bench_default_steps20.json
python bench.py > $O/r06_bench_default_noflags.json 2> $O/bench.err; cut -c1-200 $O/r06_bench_default_noflags.json
python tools/time_cli.py 1536 2>&1 | tail -4
