cd $GRAFT_REPO_ROOT; hipcc --offload-arch=gfx950 -O3 tools/ab/launch_gap.hip -o /tmp/launch_gap 2>&1 | tail -2; /tmp/launch_gap
