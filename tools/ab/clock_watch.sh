# samples the GPU clocks / power while bench.py runs (is the step clock- or power-limited?)
python bench.py --steps 30000 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode > /tmp/b.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use|busy" | tr '\n' ' ' | cut -c1-400; echo; sleep 0.3; done
wait $BP
tail -c 300 /tmp/b.json
