#!/usr/bin/env python3
"""Writes tests/golden/oracle_regression.json: digests of the CPU oracle's own outputs on seeded
inputs.  NOT reference-derived (see README.md here): it pins the oracle against silent drift between
rounds -- the anchors in survey_anchors.json pin it against the reference."""
import json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import oracle as o
from util_signals import fm_carrier, am_carrier, tones_plus_noise
import test_postchain_gpu as T


def digest(x):
    x = np.asarray(x)
    v = np.concatenate([x.real.ravel(), x.imag.ravel()]) if np.iscomplexobj(x) else x.astype(np.float64).ravel()
    w = np.cos(0.37 * np.arange(len(v)))                        # position-sensitive
    return {"n": int(len(v)), "sum": float(v.sum()), "abs": float(np.abs(v).sum()), "dot": float((v * w).sum())}


def build():
    out = {}
    fs = 2e6
    for name in ("FM", "AM", "SAM", "USB", "CWU"):
        m, kw = T.MODES[name]
        d = o.CDemodulator(2048); d.SetInputSampleRate(fs); d.SetDemod(m, T.info(o, **kw)); d.SetDemodFreq(-100e3)
        out["chain_" + name] = digest(d.process_append(T.make_input(name, 19968 * 8, fs)))
        out["chain_" + name]["smeter"] = float(d.GetSMeterAve())
    ff = o.CFastFIR(16384); ff.SetupParameters(-5000, 5000, 0, 62500.0)
    out["fastfir16384"] = digest(ff.ProcessData(tones_plus_noise(5, 16384 * 5, 62500.0, [1000.0, 20000.0])))
    dc = o.CDownConvert(); dc.SetDataRate(10e6, 15000); dc.SetFrequency(-1.2e6)
    out["downconvert_10M"] = digest(dc.ProcessData(tones_plus_noise(6, 32768, 10e6, [1.2e6 + 500.0])))
    f = o.CFft(); f.SetFFTParams(4096, False, 0.0, fs); f.SetFFTAve(3)
    for k in range(5):
        f.PutInDisplayFFT(tones_plus_noise(7 + k, 4096, fs, [250e3]))
    out["spectrum4096_ave3"] = digest(f.ave_buf())
    a = o.CAgc(); a.SetParameters(True, True, -60, 30, 5, 300, 62500.0)
    out["agc_hang"] = digest(a.ProcessData(T.level_steps(40000, 62500.0, 3)))
    r = o.CFractResampler(); r.Init(8192)
    out["resampler"] = digest(r.Resample(np.sin(0.01 * np.arange(6000)), 1.6276))
    nb = o.CNoiseProc(); nb.SetupBlanker(True, 40.0, 10.0, fs)
    x = tones_plus_noise(9, 60000, fs, [1e5]); x[::7001] += 30000.0
    out["blanker"] = digest(nb.ProcessBlanker(x))
    return out


if __name__ == "__main__":
    json.dump(build(), open(os.path.join(HERE, "oracle_regression.json"), "w"), indent=1, sort_keys=True)
    print("written")
