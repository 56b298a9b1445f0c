grep -m1 "model name" /proc/cpuinfo; nproc
for m in eager threads graph eager; do echo -n "$m: "; python bench.py --steps 400 --warmup 20 --replay $m --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | cut -c70-135; done
