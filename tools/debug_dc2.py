import sys, numpy as np, faulthandler
faulthandler.enable()
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cutesdr_amd as ca
from util_signals import tones_plus_noise
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 15
C = 6
setups = [(15000, -100e3), (10000, 40e3), (20000, -333e3), (1000, 7e3), (15000, 0.0), (10000, -1e3)]
b = ca.DownConvertBatch(C)
for c, (bw, f) in enumerate(setups):
    r = b.set_data_rate(2e6, bw, channel=c); b.set_frequency(f, channel=c)
    print("ch", c, "rate", r, "stages", b.stages(c), flush=True)
x = np.stack([tones_plus_noise(20 + c, T, 2e6, [700.0]) for c in range(C)])
print("process...", flush=True)
got = b.process(x)
print("done", [len(g) for g in got], flush=True)
