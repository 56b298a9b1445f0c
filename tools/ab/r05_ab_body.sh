cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05i; O=gpurun_out/r05i
for rep in 1 2 3 4; do for v in 1 0; do
SLIDE_BODY=$v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/body${v}_$rep.json 2> $O/body${v}_$rep.err
echo "body=$v rep $rep: $(python -c "import json;d=json.load(open('$O/body${v}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done
for rep in 1 2; do for v in 1 0; do
SLIDE_BODY=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/s20_body${v}_$rep.json 2> $O/s20_body${v}_$rep.err
echo "steps20 body=$v rep $rep: $(python -c "import json;d=json.load(open('$O/s20_body${v}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done
