#!/usr/bin/env python3
"""Builds a copy of the library whose K1 (fastfir2_kernels.hip) is compiled with extra flags, for A/B runs of
kernel experiments in one gpurun call:   python tools/altlib.py NAME [unit.hip[,unit2.hip]] [-DFLAG ...]
  -> cutesdr_amd/libcutesdr_mi_NAME.so;   then   CSDR_LIB_PATH=<that> python tools/ab_fastfir.py 0,2"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutesdr_amd import _build

name, extra = sys.argv[1], sys.argv[2:]
units = ["fastfir2_kernels"]
if extra and extra[0].endswith(".hip"):
    units, extra = [u[:-4] for u in extra[0].split(",")], extra[1:]
_build.build()
out = os.path.join(ROOT, "cutesdr_amd", "libcutesdr_mi_%s.so" % name)
alt = []
for unit in units:
    src = os.path.join(_build.CSRC, unit + ".hip")
    obj = os.path.join(_build.OBJ, "%s.%s.o" % (unit, name))
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + _build.FILE_FLAGS.get(unit + ".hip", []) + ["-I", _build.OBJ] + extra + ["-c", src, "-o", obj])
    alt.append(obj)
objs = [o for o in sorted(glob.glob(os.path.join(_build.OBJ, "*.hip.o")) + glob.glob(os.path.join(_build.OBJ, "downconv_plan_*.o")))
        if os.path.basename(o)[:-6] not in units]
subprocess.check_call([_build._hipcc(), "-shared", "-fPIC", "--offload-arch=" + _build.ARCH, "-o", out] + objs + alt)
print(out)
