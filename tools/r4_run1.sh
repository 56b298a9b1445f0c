set -x
cd /root/repo
mkdir -p gpurun_out
python tools/prec_probe.py --variants "SLIDE_GX=0/SLIDE_BODY=0/SLIDE_SA_CHAIN=0/SLIDE_ATTN_TAIL=0" > gpurun_out/probe1.log 2>&1
Q="--no-cpu-baseline --no-parity --no-roofline --no-decode"
python bench.py --steps 300 --warmup 20 $Q > gpurun_out/b_default.json 2>gpurun_out/b_default.err
python bench.py --steps 300 --warmup 20 $Q --pos-prec fp32 > gpurun_out/b_pos32.json 2>gpurun_out/b_pos32.err
python bench.py --steps 20 --warmup 5 $Q > gpurun_out/b_default20.json 2>>gpurun_out/b_default.err
python bench.py --steps 300 --warmup 20 $Q > gpurun_out/b_default2.json 2>>gpurun_out/b_default.err
tail -3 gpurun_out/probe1.log; cat gpurun_out/b_*.json | cut -c1-200
