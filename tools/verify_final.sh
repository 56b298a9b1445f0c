# final check of the committed tree (GPU box): full gpu suite, smoke, the driver's bench line, the no-flags line, the five-category line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final; O=gpurun_out/final; T=${1:-r06d}
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/${T}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_default_steps20.json 2> $O/${T}_bench20.err
python bench.py > $O/${T}_bench_default_noflags.json 2> $O/${T}_bench.err
python bench.py --steps 100 --warmup 10 --workload five-cat --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/${T}_bench_five_cat.json 2> $O/${T}_bench5.err
cat $O/${T}_gputest.log; tail -3 $O/${T}_smoke.log; cut -c1-260 $O/${T}_bench_default_steps20.json; cut -c1-200 $O/${T}_bench_default_noflags.json; cut -c1-160 $O/${T}_bench_five_cat.json
