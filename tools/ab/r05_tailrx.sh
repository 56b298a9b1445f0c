# attention tail with X fragments through registers (attn_tail_rx_kernel / rx8): parity + A/B against the ring form, same call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05t; O=gpurun_out/r05t
for v in 2 1; do
SLIDE_TAIL_RX=$v timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "golden or benched or fused or variants" > $O/test$v.log 2>&1; tail -2 $O/test$v.log
done
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline"
run() { tag=$1; shift; env "$@" python bench.py $Q > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
for r in 1 2 3; do
run rx2_$r SLIDE_TAIL_RX=2
run rx1_$r SLIDE_TAIL_RX=1
run rx0_$r SLIDE_TAIL_RX=0
done
for v in 2 1 0; do
run feat_rx$v SLIDE_TAIL_RX=$v SLIDE_BENCH_ONLY=feat
done
for v in 2 1 0; do
SLIDE_TAIL_RX=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-parity > $O/roof_rx$v.json 2> $O/roof_rx$v.err
python -c "
import json;d=json.load(open('$O/roof_rx$v.json'));r=d['roofline']
print('rx$v', d['value'], r['frac'], r['step_ms_eager_sum'], {k:v for k,v in r['mfma_kernels'].items() if 'tail' in k})"
done
