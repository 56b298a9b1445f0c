"""per-call device time and effective bandwidth of the row-major module kernels in one decode (authoring tool)"""
import os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "pointnet2"))
import torch
from slide_amd import rows as R
rec = collections.OrderedDict()
def wrap(name, fn, nbytes):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record(); torch.cuda.synchronize()
        key = (name,) + nbytes(a, k, out)[1]
        t = rec.setdefault(key, [0, 0.0, 0])
        t[0] += 1; t[1] += e0.elapsed_time(e1) * 1e3; t[2] = nbytes(a, k, out)[0]
        return out
    return w
es = lambda r: r.data.element_size()
R.norm_act = wrap("norm_act", R.norm_act, lambda a, k, o: (2 * a[0].rows * a[0].ld * es(a[0]), (a[0].rows, a[0].ld, len(a) > 1 and a[1] is not None)))
R.conv = wrap("conv", R.conv, lambda a, k, o: ((a[0].rows * a[0].ld + o.rows * o.ld) * es(o), (a[0].rows, a[0].ld, o.ld, type(a[0]).__name__)))
R.concat_qk = wrap("concat_qk", R.concat_qk, lambda a, k, o: ((a[1].rows * a[1].ld + o.rows * o.ld) * es(o), (o.rows, o.ld)))
R.attend = wrap("attend", R.attend, lambda a, k, o: (2 * a[0].rows * a[0].ld * es(o), (a[0].rows, a[0].ld, a[2])))
R._pair_conv = wrap("pair_conv", R._pair_conv, lambda a, k, o: (o.rows * o.ld * es(o), (o.rows, a[0].ld, o.ld, a[2])))
R.group = wrap("group", R.group, lambda a, k, o: (0, (o.rows, o.ld, type(o).__name__)))
R.conv_attend = wrap("conv_attend", R.conv_attend, lambda a, k, o: ((a[0].rows * a[0].ld + a[2].rows * a[2].ld + o.rows * o.ld) * es(o), (a[0].rows, a[0].ld, a[2].ld, a[3])))
exec(open(os.path.join(REPO, "tools", "time_decode.py")).read().split("for _ in range(2):")[0].replace("REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", ""))
out = ae.decode(kp, feat, label=lab); torch.cuda.synchronize(); rec.clear()
out = ae.decode(kp, feat, label=lab); torch.cuda.synchronize()
tot = sum(v[1] for v in rec.values())
for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-40s calls %2d  %8.1f us  %5.1f%%  %6.0f GB/s" % (str(k), v[0], v[1], 100 * v[1] / tot, v[2] * v[0] / v[1] / 1e3))
print("total timed %.1f ms" % (tot / 1e3))
