# position chain over a multiple of the batch, stepping once per that many rounds: tools/ab/ab_posmult.sh
cd /root/repo
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode"
run() { python bench.py $Q $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  SLIDE_POS_MULT=1 run "fp16 mult1"
  SLIDE_POS_MULT=2 run "fp16 mult2"
  SLIDE_POS_MULT=4 run "fp16 mult4"
  SLIDE_POS_MULT=1 run "possplit mult1" "--pos-prec split"
  SLIDE_POS_MULT=2 run "possplit mult2" "--pos-prec split"
  SLIDE_POS_MULT=4 run "possplit mult4" "--pos-prec split"
done
