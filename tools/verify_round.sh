# round-5 verification set (GPU box): full gpu suite, smoke, the driver's bench line, the no-flags bench line, profile refresh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final; O=gpurun_out/final
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/final_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_default_steps20.json 2> $O/final_bench20.err
python bench.py > $O/r05_bench_default_noflags.json 2> $O/final_bench.err
python bench.py --steps 300 --warmup 20 --pos-prec fp16 --no-cpu-baseline --no-decode --no-parity > $O/r05_bench_posfp16_steps300.json 2> $O/final_bench_fp16.err
bash tools/rocprof_run5.sh r05b > $O/prof.log 2>&1
cat $O/final_gputest.log; tail -3 $O/final_smoke.log; cut -c1-300 $O/r05_bench_default_steps20.json; cut -c1-200 $O/r05_bench_default_noflags.json; cut -c1-200 $O/r05_bench_posfp16_steps300.json; tail -3 $O/prof.log
