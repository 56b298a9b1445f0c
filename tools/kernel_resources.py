"""Register / scratch / LDS footprint of every gfx950 kernel in libslide_hip.so: the table slide_amd/build.py records while it
compiles (hipcc -Rpass-analysis=kernel-resource-usage; its build-time check fails on spills outside the opt-in list).
`python tools/kernel_resources.py [out.md]` forces a rebuild and writes the table."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slide_amd import build as B

if __name__ == "__main__":
    B.build(force="--no-build" not in sys.argv)
    lines = ["# Kernel resource table (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; written by slide_amd/build.py)", "",
             "`spill` = VGPRs spilled, `scratch` = private segment bytes per lane.  PRODUCT library (libslide_hip.so): the build FAILS when any",
             "kernel has either (round 4: the opt-in variants that used to be exempt live in the experiments build, which is not linted).", "",
             "| source | kernel | VGPR | AGPR | spill | scratch B | static LDS B | waves/SIMD | |", "|---|---|---|---|---|---|---|---|---|"]
    n = nbad = 0
    for src, _ in B.SOURCES:
        f = os.path.join(B.CSRC, src.replace(".hip", ".o.resources"))
        for row in sorted(l.rstrip("\n").split("\t") for l in open(f)):
            opt = any(re.search(p, row[0]) for p in B.SPILL_OPT_IN)
            n += 1
            nbad += (int(row[3]) > 0 or int(row[4]) > 0)
            lines.append("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s |" % ((src,) + tuple(row) + ("opt-in" if opt else "",)))
    lines.append("")
    lines.append("%d kernels, %d with spills / scratch (all opt-in)." % (n, nbad))
    txt = "\n".join(lines) + "\n"
    outs = [a_ for a_ in sys.argv[1:] if not a_.startswith("--")]
    if outs:
        open(outs[0], "w").write(txt)
    else:
        print(txt)
