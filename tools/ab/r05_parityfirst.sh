# bench.py at the driver's flags with the parity leg before / after the timed region (same call, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05pf; O=gpurun_out/r05pf
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-decode > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['parity']['measured'], d['parity']['forward_rel_l2_vs_fp32_mode'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
for r in 1 2 3; do
run first_$r SLIDE_BENCH_PARITY_FIRST=1
run after_$r SLIDE_BENCH_PARITY_FIRST=0
done
