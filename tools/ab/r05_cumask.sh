# the position chain on a stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask): arrangement A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05cu; O=gpurun_out/r05cu
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $Q > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'])" 2>/dev/null || tail -2 $O/$tag.err)"; }
for r in 1 2; do
run base_$r A=0
for n in 160 176 192 208 224 240; do run pos_cus${n}_$r SLIDE_POS_CUS=$n; done
done
run pos192_from64 SLIDE_POS_CUS=192 SLIDE_CU_MASK_FROM=64
run feat_cus224 SLIDE_FEAT_CUS=224
run feat_cus192_pos192from64 SLIDE_FEAT_CUS=192 SLIDE_POS_CUS=192 SLIDE_CU_MASK_FROM=0
run pos192_pm1 SLIDE_POS_CUS=192 SLIDE_POS_MULT=1
run pos128_pm1 SLIDE_POS_CUS=128 SLIDE_POS_MULT=1
run pos192_pm3 SLIDE_POS_CUS=192 SLIDE_POS_MULT=3
