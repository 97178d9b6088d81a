import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cutesdr_amd as ca
from oracle import oracle as _orc
import test_postchain_gpu as T
_orc.build(); oracle = _orc
m, kw = T.MODES["FM"]
fs = 2e6
for seedcase in range(3):
    n = 19968 * 24
    x = T.make_input("FM", n, fs) if seedcase == 0 else T.make_input("FM", n, fs) * (1.0 + 0.37 * seedcase)
    x = x.astype(np.complex64).astype(np.complex128)
    def run(mod, xin, cls):
        d = cls(2048); d.SetInputSampleRate(fs); d.SetDemod(m, T.info(mod, **kw)); d.SetDemodFreq(-100e3)
        outs = []
        for i in range(0, n, 19968):
            k, o = d.ProcessData(xin[i:i + 19968])
            if k: outs.append(np.array(o[:k]))
        return np.concatenate(outs)
    g = run(ca, x, ca.CDemodulator)
    r = run(oracle, x, oracle.CDemodulator)
    rng = np.random.default_rng(5)
    xp = (x.astype(np.complex64) * np.float32(1 + 2.0 ** -22)).astype(np.complex128)      # every sample moved by an fp32 ulp or two
    rp = run(oracle, xp, oracle.CDemodulator)
    L = min(len(g), len(r), len(rp))
    eg = T.burst_errors(g[:L], r[:L]) / T.FULL_SCALE
    ep = T.burst_errors(rp[:L], r[:L]) / T.FULL_SCALE
    print("case", seedcase, "gpu", np.array2string(eg[:8], precision=2), "pert", np.array2string(ep[:8], precision=2))
