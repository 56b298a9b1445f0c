# hardware-queue sweep (GPU_MAX_HW_QUEUES) of the default arrangement and of four / five feature sub-batches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
run() { tag=$1; shift; env "$@" python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['config']['sub_batches'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
EXTRA=""
for q in 1 3 4 5 6 7 12 16 24; do run q$q GPU_MAX_HW_QUEUES=$q; done
EXTRA="--sub-batches 4"
for q in 5 6 7 12 16; do run sb4_q$q GPU_MAX_HW_QUEUES=$q; done
EXTRA="--sub-batches 2"
for q in 3 4 8; do run sb2_q$q GPU_MAX_HW_QUEUES=$q; done
