cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $2 python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])")"; }
run "default(176)" "A=1"
run "pos 144" "SLIDE_POS_CUS=144"
run "pos 208" "SLIDE_POS_CUS=208"
run "pos all" "SLIDE_POS_CUS=0"
run "default(176)" "A=1"
run "pos 160" "SLIDE_POS_CUS=160"
run "pos 192" "SLIDE_POS_CUS=192"
