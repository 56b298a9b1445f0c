cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_pm; mkdir -p $O
for rep in 1 2; do for m in 2 3 4; do for cus in 176 128; do
SLIDE_POS_MULT=$m SLIDE_POS_CUS=$cus python bench.py --gpus 1 --steps 240 --warmup 24 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/p_${m}_${cus}_$rep.json 2>/dev/null
echo "mult $m cus $cus rep $rep: $(python -c "import json;d=json.load(open('$O/p_${m}_${cus}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done; done
