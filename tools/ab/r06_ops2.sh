cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py tests/test_hip_modules.py -x -q 2>&1 | tail -4
bash tools/ops_roofline.sh r06 > gpurun_out/r06_ops_roofline.log 2>&1
grep "ball_query" gpurun_out/r06_ops/r06_ops_roofline.md | cut -c1-200
