"""Build libcutesdr_mi.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun
snapshot.  Object files are cached per source under cutesdr_amd/_obj/ keyed on mtimes.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libcutesdr_mi.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"] + \
    os.environ.get("CSDR_EXTRA_HIPCC_FLAGS", "").split()
# per-source additions: the overlap-save kernel is a long straight-line butterfly chain with two waves
# per SIMD; LLVM's "max-ILP" machine scheduler orders it ~4 % faster than the default (measured A/B)
FILE_FLAGS = {"fastfir2_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] + os.environ.get("CSDR_K1_FLAGS", "").split(), "fastfir_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              "spectrum_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}   # K3: +4 %; K2/K4 measured slower with it


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libcutesdr_mi.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    procs = []
    for src in sources:
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    if force or procs or _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
