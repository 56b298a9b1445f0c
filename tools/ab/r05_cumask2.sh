# default arrangement with the position stream on 11/16 of the CUs: GPU tests that drive the generators + the driver's bench line, A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05cu; O=gpurun_out/r05cu
timeout 1500 python -m pytest tests/test_hip_cli.py tests/test_hip_generation.py tests/test_hip_engine.py -q -m gpu -k "cli or generat or chains or pipelin" > $O/test.log 2>&1; tail -3 $O/test.log
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-decode > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['config']['pos_stream_cus'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
for r in 1 2 3; do
run mask_$r A=0
run all_$r SLIDE_POS_CUS=0
done
