# point chain (FP0's second Mlp + head [+ DDPM update] as one launch): parity + A/B, same call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05pc; O=gpurun_out/r05pc
timeout 1500 python -m pytest tests/test_hip_engine.py tests/test_hip_cli.py -q -m gpu -x > $O/test.log 2>&1; tail -5 $O/test.log
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline"
run() { tag=$1; shift; env "$@" python bench.py $Q > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['config']['launches_per_step'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
for r in 1 2 3; do
run fused_$r SLIDE_POINT_CHAIN_UPDATE=1
run chain_$r SLIDE_POINT_CHAIN_UPDATE=0
done
run feat_fused SLIDE_POINT_CHAIN_UPDATE=1 SLIDE_BENCH_ONLY=feat
run feat_chain SLIDE_POINT_CHAIN_UPDATE=0 SLIDE_BENCH_ONLY=feat
for v in 1 0; do
SLIDE_POINT_CHAIN_UPDATE=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-parity > $O/roof_pc$v.json 2> $O/roof_pc$v.err
python -c "
import json;d=json.load(open('$O/roof_pc$v.json'));r=d['roofline']
print('upd$v', d['value'], r['step_ms_eager_sum'], {k:v for k,v in r['mfma_kernels'].items() if 'point' in k})"
done
