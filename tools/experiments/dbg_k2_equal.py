import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cutesdr_amd as ca
from util_signals import tones_plus_noise
L = ca.lib()
L.csdr__downconv_force_dynamic.restype = C.c_int; L.csdr__downconv_force_dynamic.argtypes = [C.c_int]
for in_rate, bw, chain in [(2e6, 15000, [11, 11, 15, 19, 31]), (62500, 10000, [51]), (250000, 20000, [23, 51]), (2e6, 1000, [3, 3, 11, 11, 11, 11, 15])]:
    calls = [16384, 8192 + 640, 4096 + (3 << len(chain)), 20000 - 20000 % (1 << len(chain))]
    x = tones_plus_noise(11, sum(calls), in_rate, [in_rate * 0.05 + 300.0, in_rate * 0.05 - 900.0, in_rate * 0.19])
    outs = []
    for dyn in (0, 1):
        L.csdr__downconv_force_dynamic(dyn)
        dc = ca.CDownConvert(); dc.SetDataRate(in_rate, bw); dc.SetFrequency(-in_rate * 0.05)
        pos, got = 0, []
        for n in calls:
            got.append(dc.ProcessData(x[pos:pos + n])); pos += n
        outs.append(np.concatenate(got))
    d = np.abs(outs[0] - outs[1])
    print(chain, "max diff", d.max(), "rel", d.max() / np.abs(outs[1]).max(), "n diff", int((d > 0).sum()), "of", len(d), "first", int(np.argmax(d > 0)))
L.csdr__downconv_force_dynamic(-1)
