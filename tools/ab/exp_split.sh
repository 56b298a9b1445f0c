run() { echo -n "$1: "; env $1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launches_per_step'])"; }
run "A=1"
run "SLIDE_SPLIT_FIRST=256"
run "SLIDE_SPLIT_FIRST=32"
run "SLIDE_ATTN_TAIL=0"
run "SLIDE_FUSE_FIN=0"
run "A=1"
