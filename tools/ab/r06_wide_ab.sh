cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_wide; mkdir -p $O
for rep in 1 2 3; do for v in 0 1; do
SLIDE_POINT_CHAIN_WIDE=$v python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/w_${v}_$rep.json 2>/dev/null
echo "wide $v rep $rep: $(python -c "import json;d=json.load(open('$O/w_${v}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done
