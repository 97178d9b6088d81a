#!/usr/bin/env python3
"""Exploration (GPU box): error of the chain against the oracle from sample 0, per FastFIR burst, mono and
stereo, every mode -- the data behind the bounds in tests/test_chain_parity_gpu.py."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cutesdr_amd as ca
from oracle import oracle
from test_postchain_gpu import MODES, info, make_input
from util_signals import FULL_SCALE
MODES = dict(MODES)
MODES["CWL"] = (6, dict(HiCut=500, LowCut=-500, HiCutmin=50, HiCutmax=1000, LowCutmax=-50, LowCutmin=-1000, Offset=700, Symetric=0))
def inp(mode, n, fs):
    if mode == "CWL":
        from util_signals import tones_plus_noise
        return tones_plus_noise(9, n, fs, [100e3, 100e3 - 300.0])
    return make_input(mode, n, fs)
fs = 2e6
res = {}
for mode in MODES:
    m, kw = MODES[mode]
    for stereo in (False, True):
        d, r = ca.CDemodulator(2048), oracle.CDemodulator(2048)
        for obj, mod in ((d, ca), (r, oracle)):
            obj.SetInputSampleRate(fs); obj.SetDemod(m, info(mod, **kw)); obj.SetDemodFreq(-100e3)
        lim = d.buf_limit()
        n = lim * 40
        x = inp(mode, n, fs)
        errs, amps = [], []
        for i in range(0, n, lim):
            kg, og = d.ProcessData(x[i:i + lim], stereo)
            kr, orr = r.ProcessData(x[i:i + lim], stereo)
            assert kg == kr
            for j in range(0, kr, 1024):
                errs.append(float(np.abs(og[j:j + 1024] - orr[j:j + 1024]).max() / FULL_SCALE))
                amps.append(float(np.abs(orr[j:j + 1024]).max() / FULL_SCALE))
        res["%s%s" % (mode, "/st" if stereo else "")] = {"bursts": len(errs), "err": ["%.1e" % e for e in errs[:12]], "max_all": max(errs),
                                                         "amp": ["%.2f" % a for a in amps[:12]]}
for k, v in res.items():
    print(k, v["bursts"], "max %.2e" % v["max_all"], v["err"], v["amp"])
