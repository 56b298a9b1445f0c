cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $2 python bench.py --gpus 1 --steps 240 --warmup 24 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d['config'].get('pos_batch_multiple'))")"; }
run "mult 2" "SLIDE_POS_MULT=2"
run "mult 3" "SLIDE_POS_MULT=3"
run "mult 4" "SLIDE_POS_MULT=4"
run "mult 1" "SLIDE_POS_MULT=1"
run "mult 2" "SLIDE_POS_MULT=2"
