#!/usr/bin/env python3
"""VGPRs, LDS, scratch and SGPRs of every kernel in the built library (from the code objects' metadata notes):
   python tools/kernel_resources.py [regex]"""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.environ.get("CSDR_LIB_PATH") or os.path.join(ROOT, "cutesdr_amd", "libcutesdr_mi.so")
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
tmp = tempfile.mkdtemp()
shutil.copy(lib, os.path.join(tmp, "lib.so"))
subprocess.run([LLVM + "/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, capture_output=True)
rows = set()
for f in glob.glob(os.path.join(tmp, "lib.so.*gfx950*")):
    out = subprocess.run([LLVM + "/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
    for blk in out.split("- .agpr_count")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        if g("name"):
            rows.add((g("name").group(1), int(g("vgpr_count").group(1)), int(g("group_segment_fixed_size").group(1)),
                      int(g("private_segment_fixed_size").group(1)), int(g("sgpr_count").group(1))))
shutil.rmtree(tmp)
names = sorted(rows)
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in names), capture_output=True, text=True).stdout.split("\n")
print("vgpr  lds_static scratch sgpr  kernel")
for (n, v, l, p, s), d in zip(names, dem):
    if pat.search(d):
        print("%4d %10d %7d %4d  %s" % (v, l, p, s, d[:140]))
