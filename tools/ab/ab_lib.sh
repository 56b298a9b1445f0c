# A/B of two builds of the product library in bench.py's arrangement: tools/ab/ab_lib.sh <libA.so> <libB.so> [rounds]
cd /root/repo
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode"
run() { SLIDE_HIP_LIB=$1 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in $(seq 1 ${3:-3}); do echo "A $(run $1)"; echo "B $(run $2)"; done
