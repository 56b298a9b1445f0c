cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_prec; mkdir -p $O
python tools/prec_probe.py --nets feat --batch 64 --variants "SLIDE_POINT_CHAIN_WIDE=0/SLIDE_POINT_CHAIN=0" > $O/probe2.txt 2>&1
grep -h "fp16\|launches" $O/probe2.txt | grep -v "rounded W" | cut -c1-330
python -m pytest tests/test_hip_engine.py -k "benched_arithmetic or point_chain or denoiser_forward" -s -q 2>&1 | grep -v "^$" | tail -25
python -m pytest tests/test_hip_modules.py -k "decode" -s -q 2>&1 | grep "fp16-operand\|passed\|failed\|Error" | tail
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12
