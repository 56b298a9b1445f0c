cd /root/repo; mkdir -p gpurun_out
( for inp in normal keypoints scaled:0.3 scaled:3; do for t in "" 0 500 999; do
python tools/prec_probe.py --nets pos --input $inp --t "$t" 2>&1 | grep -v amdgpu.ids
done; done ) > gpurun_out/probe2.log 2>&1
