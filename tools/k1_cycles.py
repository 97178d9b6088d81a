#!/usr/bin/env python3
"""Shader cycles per block and the clock the chip holds under K1, from a diagnostic build (-DK1_CYC: one
s_memtime / s_memrealtime pair around the block loop of every workgroup).  Separates "fewer cycles" from
"higher clock" when two builds are compared (the chip lowers its clock under load, MI355X_MICROARCH.md DVFS).
   python tools/altlib.py NAME -DK1_CYC [flags]  ;  CSDR_LIB_PATH=.../libcutesdr_mi_NAME.so python tools/k1_cycles.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
Cn, T = 256, 1 << 19
dev = torch.device("cuda", 0)
x = torch.randn((Cn, T, 2), device=dev, dtype=torch.float32) * 3276.7
y = torch.empty_like(x)
dbg = torch.zeros((Cn * 2,), device=dev, dtype=torch.int64)
L = ca.lib()
L.csdr__dbg_fastfir_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
ff = ca.FastFirBatch(Cn, 16384); ff.setup(-5000, 5000, 0, 62500.0)
st = torch.cuda.current_stream().cuda_stream
assert L.csdr__dbg_fastfir_stage(ff.h, 0, C.c_void_p(dbg.data_ptr())) == 0
for _ in range(300): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
e1.record(); torch.cuda.synchronize()
a = dbg.view(Cn, 2).double().cpu().numpy()
cyc, rt = a[:, 0], a[:, 1]
import numpy as np
print(json.dumps({"ms": round(e0.elapsed_time(e1) / 100, 4), "cycles_per_block_median": round(float(np.median(cyc)) / 64, 1),
                  "cycles_per_block_max": round(float(cyc.max()) / 64, 1),
                  "clock_GHz_median": round(float(np.median(cyc / rt)) * 0.1, 3), "loop_us_median": round(float(np.median(rt)) / 100, 1)}))
