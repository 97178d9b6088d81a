"""Exploration (GPU box): where the first FM burst of the chain departs from the oracle -- stage by stage against
the oracle taps.  Result: down-converter, filter and AGC agree to 1e-7 of full scale; the FM stage fed with the
oracle stream agrees to 1e-7 too; fed with the fp32 chain it differs at the very first non-zero samples
(amplitude ~1e-7 of full scale: the phase of rounding noise).  Basis of tests/test_chain_parity_gpu.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cutesdr_amd as ca
from oracle import oracle
from test_postchain_gpu import MODES, info, make_input
from util_signals import FULL_SCALE
fs = 2e6
m, kw = MODES["FM"]
r = oracle.CDemodulator(2048); d = ca.CDemodulator(2048)
for obj, mod in ((d, ca), (r, oracle)):
    obj.SetInputSampleRate(fs); obj.SetDemod(m, info(mod, **kw)); obj.SetDemodFreq(-100e3)
r.enable_taps(True)
lim = r.buf_limit()
x = make_input("FM", lim * 6, fs)
# GPU leaves chained by hand
dc = ca.CDownConvert(); dc.SetDataRate(fs, 15000.0); dc.SetCwOffset(0.0); dc.SetFrequency(-100e3)
ff = ca.CFastFIR(2048); ff.SetupParameters(-5000, 5000, 0, 62500.0)
ag = ca.CAgc(); ag.SetParameters(True, False, -100, 30, 0, 200, 62500.0)
fm = ca.CFmDemod(62500.0); fm.SetSquelch(0)
for i in range(0, len(x), lim):
    r.clear_taps()
    k, o = r.ProcessData(x[i:i + lim])
    kg, og = d.ProcessData(x[i:i + lim])
    y1 = dc.ProcessData(x[i:i + lim]); y2 = ff.ProcessData(y1)
    t1 = r.tap(1)
    print("call", i // lim, "k", k, kg, "dc err %.2e" % (np.abs(y1 - t1).max() / FULL_SCALE), end=" ")
    if len(y2):
        t2, t3, t4 = r.tap(2), r.tap(3), r.tap(4)
        y3 = ag.ProcessData(y2); y4 = fm.ProcessData(y3, 5000.0)
        e4 = np.abs(y4 - t4); e4c = np.abs(og[:k] - o[:k])
        print("fir %.2e agc %.2e fm(leaves) %.2e at %d  fm(chain) %.2e at %d | audio[%d] want %.1f leaf %.1f chain %.1f" % (
            np.abs(y2 - t2).max() / FULL_SCALE, np.abs(y3 - t3).max() / FULL_SCALE, e4.max() / FULL_SCALE, int(np.argmax(e4)),
            e4c.max() / FULL_SCALE, int(np.argmax(e4c)), int(np.argmax(e4c)), o[int(np.argmax(e4c))], y4[int(np.argmax(e4c))], og[int(np.argmax(e4c))]))
    else:
        print()
