# generated-X GEMM tile widths after the round-6 kernel changes: 64-channel tiles (default) against 128-channel tiles
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $2 python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])")"; }
run "default" "A=1"
run "N64W=0" "SLIDE_GX_N64W=0"
run "N64=0" "SLIDE_GX_N64=0"
run "N64=0 N64W=0" "SLIDE_GX_N64=0 SLIDE_GX_N64W=0"
run "DUAL=0" "SLIDE_GX_DUAL=0"
run "default" "A=1"
for v in "A=1" "SLIDE_GX_N64W=0"; do echo "== $v"; env $v python tools/profile_ops.py --batch 88 2>&1 | grep -v amdgpu | awk '{print $1, $2, $3}' | tr '\n' ';'; echo; done
