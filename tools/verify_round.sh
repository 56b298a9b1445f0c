# round-6 (second session) verification set (GPU box): full gpu suite, smoke, the driver's bench line, the no-flags bench line, the
# chunk-major A/B, the profile set r06b (kernel trace, FETCH / WRITE / MFMA counters, batch curve), per-workgroup timelines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final; O=gpurun_out/final
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/final_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06b_bench_default_steps20.json 2> $O/final_bench20.err
python bench.py > $O/r06b_bench_default_noflags.json 2> $O/final_bench.err
for v in 0 1; do SLIDE_FM=$v SLIDE_PACKED_VECS=$v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/r06b_bench_fm_pv_$v.json 2> $O/final_bench_ab$v.err; done
python bench.py --steps 100 --warmup 10 --workload five-cat --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/r06b_bench_five_cat.json 2> $O/final_bench5.err
( export SLIDE_CHAIN_P=0 SLIDE_GX_DUAL=0; for b in 88 688; do for k in kind16 kind17 kind19 kind31 kind1; do python tools/ab/op_timeline.py feat $b $k 2>&1 | grep -v amdgpu.ids; done; done ) > $O/r06b_timelines.txt 2>&1
bash tools/rocprof_run6.sh r06b > $O/prof.log 2>&1
cat $O/final_gputest.log; tail -3 $O/final_smoke.log; cut -c1-300 $O/r06b_bench_default_steps20.json; cut -c1-200 $O/r06b_bench_default_noflags.json; cut -c1-120 $O/r06b_bench_fm_pv_0.json; cut -c1-120 $O/r06b_bench_fm_pv_1.json; tail -3 $O/prof.log
