cd $GRAFT_REPO_ROOT; hipcc --offload-arch=gfx950 -O3 tools/ab/launch_gap.hip -o /tmp/launch_gap 2>/dev/null
for q in 4 5 6 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q /tmp/launch_gap $1 | grep "wgs  64 spin 40000"; done
