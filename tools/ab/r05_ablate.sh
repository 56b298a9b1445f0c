# TIMING ablations of the sampling arrangement (results are wrong by construction): what would the step cost if a class of launches were free?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05abl; O=gpurun_out/r05abl
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline"
run() { tag=$1; shift; env "$@" python bench.py $Q > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['config']['launches_per_step'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
run base A=0
run no_gemm16 SLIDE_ABL_DROP=gemm16
run no_pair_first SLIDE_ABL_DROP=pair_first
run no_16row SLIDE_ABL_DROP=gemm16,pair_first,pair_norm2
run no_tail SLIDE_ABL_DROP=tail
run no_gx SLIDE_ABL_DROP=gemm_gx
run no_sa_chain SLIDE_ABL_DROP=sa_chain
run base2 A=0
run feat_base SLIDE_BENCH_ONLY=feat
run feat_no_16row SLIDE_BENCH_ONLY=feat SLIDE_ABL_DROP=gemm16,pair_first
run feat_no_tail SLIDE_BENCH_ONLY=feat SLIDE_ABL_DROP=tail
run feat_no_gx SLIDE_BENCH_ONLY=feat SLIDE_ABL_DROP=gemm_gx
run feat_no_sa SLIDE_BENCH_ONLY=feat SLIDE_ABL_DROP=sa_chain
