#!/usr/bin/env python3
"""Where a block of K1 spends its cycles: builds a diagnostic copy of the library with in-kernel stamps
(-DCSDR_K1_STAMPS, fastfir2_kernels.hip), runs the C3 workload and prints the share of every pass.
  step 1 (here, no GPU):  python tools/k1_stamps.py build     -> cutesdr_amd/libcutesdr_mi_stamps.so
  step 2 (GPU box):       CSDR_LIB_PATH=cutesdr_amd/libcutesdr_mi_stamps.so python tools/k1_stamps.py run
The stamped build's own run time means nothing (its fences forbid overlaps the real kernel has): read shares."""
import ctypes as C, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "cutesdr_amd", "libcutesdr_mi_stamps.so")

def build():
    from cutesdr_amd import _build
    _build.build()
    src = os.path.join(_build.CSRC, "fastfir2_kernels.hip")
    obj = os.path.join(_build.OBJ, "fastfir2_kernels.stamps.o")
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + _build.FILE_FLAGS["fastfir2_kernels.hip"] +
                          ["-DCSDR_K1_STAMPS", "-c", src, "-o", obj])
    objs = [o for o in sorted(glob.glob(os.path.join(_build.OBJ, "*.hip.o")) + glob.glob(os.path.join(_build.OBJ, "downconv_plan_*.o")))
            if not o.endswith("fastfir2_kernels.hip.o")]
    subprocess.check_call([_build._hipcc(), "-shared", "-fPIC", "--offload-arch=" + _build.ARCH, "-o", OUT] + objs + [obj])
    print(OUT)

def run():
    import torch
    import cutesdr_amd as ca
    Cn, T = 256, 1 << 19
    dev = torch.device("cuda", 0)
    x = torch.randn((Cn, T, 2), device=dev, dtype=torch.float32) * 3276.7
    y = torch.empty_like(x)
    dbg = torch.zeros((Cn * 8 * 16,), device=dev, dtype=torch.int64)
    L = ca.lib()
    L.csdr__fastfir_set_variant.argtypes = [C.c_void_p, C.c_int]
    L.csdr__dbg_fastfir_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    ff = ca.FastFirBatch(Cn, 16384); ff.setup(-5000, 5000, 0, 62500.0)
    assert L.csdr__fastfir_set_variant(ff.h, int("2")) == 0
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(50): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
    assert L.csdr__dbg_fastfir_stage(ff.h, 0, C.c_void_p(dbg.data_ptr())) == 0
    ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
    torch.cuda.synchronize()
    a = dbg.view(Cn, 8, 16).double().cpu().numpy()          # [workgroup][wave][segment] cycles over 64 blocks
    names = ["F1", "barrier1", "F2 tail", "F3+H+I1", "I2", "barrier2", "I3", "F2 heads", "F2 middle+Hissue"]
    tot = a.sum(axis=2)
    res = {"cycles_per_block_per_wave": round(float(tot.mean()) / 64, 1)}
    for i, n in enumerate(names):
        res[n] = {"share": round(float(a[:, :, i].sum() / tot.sum()), 4), "cycles_per_block": round(float(a[:, :, i].mean()) / 64, 1),
                  "by_wave": [round(float(a[:, w, i].mean()) / 64, 1) for w in range(8)]}
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
