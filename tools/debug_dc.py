import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cutesdr_amd as ca
from oracle import oracle
from util_signals import tones_plus_noise
np.set_printoptions(linewidth=200, precision=4, suppress=True)
dc = ca.CDownConvert(); ref = oracle.CDownConvert()
dc.SetDataRate(2e6, 15000); ref.SetDataRate(2e6, 15000)
dc.SetFrequency(50e3); ref.SetFrequency(50e3)
x = tones_plus_noise(5, 512 * 8, 2e6, [-50e3 + 2000.0, 300e3])
for i in range(8):
    got = dc.ProcessData(x[i * 512:(i + 1) * 512]); want = ref.ProcessData(x[i * 512:(i + 1) * 512])
    print(i, "err", np.abs(got - want).round(3))
