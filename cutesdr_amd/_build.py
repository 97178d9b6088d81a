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


# Decimator plans the down-converter is compiled for (downconv_plan.hip, one object each): EVERY stage sequence
# CDownConvert::SetDataRate (dsp/downconvert.cpp:127-166) can produce.  A stage's kind depends only on rho = rate at
# that stage / bandwidth (CIC-3 above 1/CIC3_MAXBW, else the first half band whose threshold it reaches,
# dsp/filtercoef.h:17-28), the rate halves from stage to stage, and the selection stops below the last threshold, at
# the 15.8 kHz floor or after nine stages: so the sequences are the prefixes of one sequence per interval between
# consecutive break points threshold * 2^s of the input ratio -- 164 of them.  all_dc_plans() also returns, for
# every plan, a (rate, bandwidth) pair that selects it (the tests run each one); tools/list_dc_plans.py prints the
# ones behind the reference's radios.  The kernel with a run-time plan stays as the form they are checked against.
def _difference(text):
    """'(.5-.475)' -> 0.5 - 0.475: the thresholds are written as differences in dsp/filtercoef.h:17-28 and kept so"""
    import re
    m = re.fullmatch(r"\(?\s*([0-9.]+)\s*-\s*([0-9.]+)\s*\)?", text.strip())
    if not m:
        raise ValueError("unexpected threshold expression in csdr_hb_taps.h: %r" % text)
    return float(m.group(1)) - float(m.group(2))


def _hb_tables():
    import re
    h = open(os.path.join(HERE, "..", "include", "csdr_hb_taps.h")).read()
    maxbw = [_difference(x) for x in re.search(r"csdr_hb_maxbw\[[^\]]*\]\s*=\s*\{([^}]*)\}", h).group(1).split(",") if x.strip()]
    lens = [int(x) for x in re.search(r"csdr_hb_len\[[^\]]*\]\s*=\s*\{([^}]*)\}", h).group(1).split(",") if x.strip()]
    cic3 = _difference(re.search(r"#define CSDR_CIC3_MAXBW\s+(\S+)", h).group(1))
    return maxbw, lens, cic3


def dc_plan(rate, bw, tables=None):
    """the stage kinds SetDataRate picks (dc_host.hpp: dc_make_plan)"""
    maxbw, lens, cic3 = tables or _hb_tables()
    f, k = rate, []
    while bw > 0 and f > bw / maxbw[-1] and f > 7900.0 * 2.0 and len(k) < 9:
        if f >= bw / cic3:
            k.append(3)
        else:
            k.append(next(lens[i] for i, m in enumerate(maxbw) if f >= bw / m))
        f /= 2.0
    return tuple(k)


def all_dc_plans():
    """{plan: (rate, bandwidth) that selects it} for every plan there is"""
    tables = _hb_tables()
    maxbw, lens, cic3 = tables
    thr = [1.0 / cic3] + [1.0 / x for x in maxbw]
    bps = sorted({t * 2.0 ** s for t in thr for s in range(12)})
    mids = [(a * b) ** 0.5 for a, b in zip(bps, bps[1:])] + [bps[-1] * 1.01]
    out = {}
    for r in mids:
        full = dc_plan(15800.0 * 2.0 ** 12, 15800.0 * 2.0 ** 12 / r, tables)
        for n in range(1, len(full) + 1):
            if full[:n] in out:
                continue
            rate = 15800.0 * 2.0 ** n if n < len(full) else 15800.0 * 2.0 ** (n + 2)      # the floor cuts a prefix
            assert dc_plan(rate, rate / r, tables) == full[:n]
            out[full[:n]] = (rate, rate / r)
    return out


# The front-end rates of the reference's radios (interface/sdrinterface.cpp:75-114: SDR-IQ, NetSDR, SDR-IP), other
# common front-end rates incl. the BASELINE configurations (2 and 10 MS/s) and the chain's own decimated rates, and
# the demodulators' maximum bandwidths (gui/mainwindow.cpp:1006-1050: CW 1 kHz, AM/SAM 10 kHz, FM 15 kHz, SSB 20 kHz).
RADIO_RATES = [66666666.6667 / d for d in (1200.0, 600.0, 420.0, 340.0)] + [80.0e6 / d for d in (1280.0, 320.0, 128.0, 130.0, 40.0)]
MORE_RATES = [62500.0, 125000.0, 250000.0, 500000.0, 625000.0, 1.0e6, 1.024e6, 2.0e6, 2.048e6, 2.4e6, 2.5e6, 3.2e6, 8e6, 10e6]
DEMOD_BWS = [1000.0, 10000.0, 15000.0, 20000.0]


def default_dc_plans():
    """the plans those rates x bandwidths select: what the library is compiled for by default"""
    tables = _hb_tables()
    out = {}
    for r in RADIO_RATES + MORE_RATES:
        for b in DEMOD_BWS:
            p = dc_plan(r, b, tables)
            if p:
                out.setdefault(p, (r, b))
    return out


# Compiled by default: the plans behind the rates above (a few dozen).  CSDR_ALL_DC_PLANS=1 compiles all 164 the
# selection rule can produce (~2 min on 8 cores, a 7 MB library); any other plan runs in the same kernel with the plan
# taken at run time (downconv_kernel<DcPlanDyn>: same words, 1.5x the time).
ALL_PLANS = os.environ.get("CSDR_ALL_DC_PLANS", "0") not in ("", "0")
DC_PLAN_PAIRS = all_dc_plans() if ALL_PLANS else default_dc_plans()
DC_PLANS = sorted(DC_PLAN_PAIRS, key=lambda p: (len(p), p))
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


# CPU-side sanitizer builds of the library's HOST code (tools/sanitize_host.sh): the same sources and device code,
# host halves instrumented (-fno-gpu-sanitize: GPU AddressSanitizer needs xnack+ code objects, which this pool does
# not run), into cutesdr_amd/_san/<kind>/ -- never the library the product loads.
SANITIZERS = {"asan": ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-gpu-sanitize"],
              "tsan": ["-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-fno-gpu-sanitize"]}


def build(force=False, verbose=False, sanitize=None, variant=None, variant_flags=()):
    """variant=NAME + variant_flags: a diagnostic copy of the whole library compiled with extra flags (every unit, the
    down-converter plans included) into cutesdr_amd/_var/NAME/libcutesdr_mi_NAME.so -- never the library the product
    loads; pick it with CSDR_LIB_PATH (tools/wg_trace.py builds its traced copy this way)."""
    global OBJ, LIB, FLAGS
    if variant:
        saved = (OBJ, LIB, FLAGS)
        OBJ = os.path.join(HERE, "_var", variant)
        LIB = os.path.join(OBJ, "libcutesdr_mi_%s.so" % variant)
        FLAGS = FLAGS + list(variant_flags)
        try:
            return _build(force, verbose)
        finally:
            OBJ, LIB, FLAGS = saved
    if sanitize:
        saved = (OBJ, LIB, FLAGS)
        OBJ = os.path.join(HERE, "_san", sanitize)
        LIB = os.path.join(OBJ, "libcutesdr_mi_%s.so" % sanitize)
        FLAGS = [f for f in FLAGS if f != "-O3"] + SANITIZERS[sanitize]
        try:
            return _build(force, verbose, link_extra=SANITIZERS[sanitize][2:3])
        finally:
            OBJ, LIB, FLAGS = saved
    return _build(force, verbose)


def _build(force=False, verbose=False, link_extra=()):
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
        stamp = obj + ".flags"                           # an object depends on every flag of its compile command
        flags = " ".join(FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + extra)
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
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + list(link_extra) + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    kind = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--sanitize=")), None)
    var = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")), None)
    print(build(force="--force" in sys.argv, verbose=True, sanitize=kind, variant=var,
                variant_flags=[a for a in sys.argv[1:] if a.startswith("-D")]))
