# arrangement experiment: the position chain on the queue of a (smaller) fourth feature sub-batch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05sh; O=gpurun_out/r05sh
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline"
run() { tag=$1; shift; env "$@" python bench.py $Q $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['config']['sub_batches'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
EXTRA=""; run base A=0
EXTRA=""; run sub3_share2 SLIDE_POS_SHARE=2
EXTRA="--sub-batches 4"; run sub4_own A=0
EXTRA="--sub-batches 4"; run sub4_share3 SLIDE_POS_SHARE=3
EXTRA="--sub-batches 4"; run sub4_72_40_share3 SLIDE_POS_SHARE=3 SLIDE_SUB_SIZES=72,72,72,40
EXTRA="--sub-batches 4"; run sub4_76_28_share3 SLIDE_POS_SHARE=3 SLIDE_SUB_SIZES=76,76,76,28
EXTRA="--sub-batches 4"; run sub4_80_16_share3 SLIDE_POS_SHARE=3 SLIDE_SUB_SIZES=80,80,80,16
EXTRA="--sub-batches 4"; run sub4_72_40_share3_pm1 SLIDE_POS_SHARE=3 SLIDE_SUB_SIZES=72,72,72,40 SLIDE_POS_MULT=1
EXTRA=""; run base2 A=0
