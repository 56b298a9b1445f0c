cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
run() { tag=$1; shift; env "$@" python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['config']['sub_batches'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
for sb in 2 3 4; do EXTRA="--sub-batches $sb"; run featonly_sb$sb SLIDE_BENCH_ONLY=feat; done
EXTRA="--sub-batches 4"; run featonly_sb4_sizes SLIDE_BENCH_ONLY=feat SLIDE_SUB_SIZES=72,64,64,56
EXTRA=""; run posonly SLIDE_BENCH_ONLY=pos
