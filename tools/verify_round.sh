# round-6 (third session) verification set (GPU box): full gpu suite, smoke, the driver's bench line, the no-flags bench line, the
# five-category shard, the A/B against the build before this session's occupancy changes (build_tmp/libbase0.so when present), the
# profile set r06c (kernel trace, FETCH / WRITE / MFMA counters, batch curve), named per-op times of one feature step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final; O=gpurun_out/final; T=${1:-r06c}
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/final_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_default_steps20.json 2> $O/final_bench20.err
python bench.py > $O/${T}_bench_default_noflags.json 2> $O/final_bench.err
python bench.py --steps 100 --warmup 10 --workload five-cat --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/${T}_bench_five_cat.json 2> $O/final_bench5.err
if [ -f build_tmp/libbase0.so ]; then bash tools/ab/r06_libab.sh base0 3 > $O/${T}_ab_base0.txt 2>&1; fi
for b in 88 688; do python tools/profile_ops.py --batch $b 2>&1 | grep -v amdgpu.ids > $O/${T}_ops_$b.txt; done
bash tools/rocprof_run6.sh $T > $O/prof.log 2>&1
cat $O/final_gputest.log; tail -3 $O/final_smoke.log; cut -c1-300 $O/${T}_bench_default_steps20.json; cut -c1-200 $O/${T}_bench_default_noflags.json; cat $O/${T}_ab_base0.txt; tail -3 $O/prof.log
