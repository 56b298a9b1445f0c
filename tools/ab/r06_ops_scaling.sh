# per-op device time of a feature / position step alone on the GPU, at several samples per launch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_ops; mkdir -p $O
for b in 88 344 688; do python tools/profile_ops.py --which feat --prec fp16 --batch $b > $O/feat_$b.txt 2>&1; done
for b in 512 2048; do python tools/profile_ops.py --which pos --prec split --batch $b > $O/pos_$b.txt 2>&1; done
tail -3 $O/*.txt
