cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -x -q 2>&1 | tail -4
bash tools/ops_roofline.sh r06 > gpurun_out/r06_ops_roofline.log 2>&1
grep "ball_query\|three_nn\|knn_points\|gather_points" gpurun_out/r06_ops/r06_ops_roofline.md | cut -c1-260
