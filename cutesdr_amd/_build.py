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


# Decimator plans the down-converter is compiled for (downconv_plan.hip, one object each): the stage sequences
# CDownConvert::SetDataRate picks for the reference's radio rates x demodulator bandwidths -- tools/list_dc_plans.py
# prints this list with the (rate / bandwidth) pairs behind every line.  Any other sequence runs the same kernel
# with a run-time plan.
DC_PLANS = [
    (23,), (27,), (35,), (39,), (51,),
    (11, 15), (19, 27), (19, 31), (19, 35), (23, 43), (23, 51),
    (11, 11, 15), (11, 15, 27), (15, 19, 35), (15, 23, 51),
    (11, 11, 11, 15), (11, 11, 15, 19), (11, 15, 19, 35),
    (11, 11, 15, 19, 31), (11, 11, 15, 23, 51),
    (11, 11, 11, 11, 15, 19), (11, 11, 11, 15, 23, 51),
    (3, 3, 11, 11, 11, 11, 15),
    # the same bandwidths at other common front-end rates (1.024 / 2.048 / 2.4 / 2.5 / 3.2 / 8 / 10 MS/s; SURVEY's
    # config C5 is 10 MS/s FM): tools/list_dc_plans.py --more
    (11, 15, 19, 31), (11, 15, 23, 47),
    (11, 11, 11, 15, 27), (11, 11, 11, 19, 27), (11, 11, 15, 19, 35), (11, 11, 15, 23, 47),
    (11, 11, 11, 11, 19, 27), (11, 11, 11, 15, 19, 35), (11, 11, 11, 15, 23, 43), (11, 11, 11, 15, 23, 47),
    (3, 11, 11, 11, 11, 15, 19), (3, 11, 11, 11, 11, 15, 27), (11, 11, 11, 11, 15, 19, 31),
    (11, 11, 11, 11, 15, 19, 35), (11, 11, 11, 11, 15, 23, 51),
    (3, 3, 3, 11, 11, 11, 11, 15), (3, 3, 11, 11, 11, 11, 15, 19), (3, 11, 11, 11, 11, 15, 19, 35),
    (3, 11, 11, 11, 11, 15, 23, 51),
    (3, 3, 3, 3, 11, 11, 11, 11, 15),
]
JOBS = max(1, min(8, os.cpu_count() or 1))


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
    # the table of precompiled down-converter plans, next to the objects (rewritten only when it changes)
    inc = os.path.join(OBJ, "downconv_plans.inc")
    text = "".join("DC_PLAN(%d, %s)\n" % (i, ", ".join(map(str, k))) for i, k in enumerate(DC_PLANS))
    if not os.path.exists(inc) or open(inc).read() != text:
        with open(inc, "w") as f:
            f.write(text)
    jobs = []                                            # (source, object, extra flags, extra deps)
    for src in sources:
        jobs.append((src, os.path.join(OBJ, os.path.basename(src) + ".o"), ["-I", OBJ], [inc]))
    plan_src = os.path.join(CSRC, "downconv_plan.hip")
    for i, k in enumerate(DC_PLANS):
        jobs.append((plan_src, os.path.join(OBJ, "downconv_plan_%d.o" % i),
                     ["-DDC_PLAN_ID=%d" % i, "-DDC_PLAN_KINDS=" + ",".join(map(str, k))], []))
    for stale in glob.glob(os.path.join(OBJ, "downconv_plan_*.o")):
        if stale not in [j[1] for j in jobs]:
            os.remove(stale)
    objs = [j[1] for j in jobs]
    todo = []
    for src, obj, extra, deps in jobs:
        stamp = obj + ".flags"                           # a plan object depends on its -D flags too
        flags = " ".join(extra)
        if force or _newer(obj, [src] + headers + deps) or not os.path.exists(stamp) or open(stamp).read() != flags:
            todo.append((src, obj, extra, stamp, flags))
    procs = []
    running = []

    def reap(block_until_below):
        while len(running) > block_until_below:
            src, obj, stamp, flags, p = running.pop(0)
            out, _ = p.communicate()
            if p.returncode != 0:
                for r in running:
                    r[4].kill()
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
            with open(stamp, "w") as f:
                f.write(flags)

    for src, obj, extra, stamp, flags in todo:
        reap(JOBS - 1)
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        running.append((src, obj, stamp, flags, p))
        procs.append(obj)
    reap(0)
    if force or procs or _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
