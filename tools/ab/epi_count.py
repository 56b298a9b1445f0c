import os, sys, runpy
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from slide_amd import rows
orig = rows._ConvPlan.epi
stat = {"calls": 0, "miss": 0, "clear": 0}
def epi(self, out, stats=None, pre_relu=False, pre_add=None, out_f32=False):
    n0 = len(self.epis)
    stat["calls"] += 1
    e = orig(self, out, stats, pre_relu, pre_add, out_f32)
    if len(self.epis) != n0: stat["miss"] += 1
    if len(self.epis) < n0: stat["clear"] += 1
    return e
rows._ConvPlan.epi = epi
runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "time_decode.py"), run_name="__main__")
print(stat)
