cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_prec; mkdir -p $O
python tools/prec_probe.py --nets feat --batch 64 --variants "SLIDE_POINT_CHAIN=0/SLIDE_SA_CHAIN=0/SLIDE_ATTN_TAIL=0/SLIDE_TAIL_RX=0/SLIDE_CHAIN_P=0" > $O/probe_default.txt 2>&1
SLIDE_EXPERIMENTS=1 python tools/prec_probe.py --nets feat --batch 64 --variants "SLIDE_GX=0/SLIDE_PAIR_FUSED=0" > $O/probe_exp.txt 2>&1
python -m pytest tests/test_hip_modules.py -k "decode" -s -q > $O/decode_test.txt 2>&1
grep -h "fp16\|launches" $O/probe_default.txt $O/probe_exp.txt | grep -v "rounded W" | cut -c1-400
grep -h "decode\|passed\|failed" $O/decode_test.txt | tail
