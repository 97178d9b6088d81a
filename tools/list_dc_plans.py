#!/usr/bin/env python3
"""The decimator plans (stage sequences of CDownConvert::SetDataRate, dsp/downconvert.cpp:127-166) that the
reference's radios can ask for: its sample-rate tables (interface/sdrinterface.cpp:75-114: SDR-IQ, NetSDR, SDR-IP)
x the demodulators' maximum bandwidths (gui/mainwindow.cpp:1006-1050: AM/SAM 10 kHz, FM 15 kHz, SSB 20 kHz,
CW 1 kHz).  cutesdr_amd/_build.py compiles the down-converter once per plan listed in DC_PLANS (a kernel that knows
its stage sequence at compile time); every other sequence runs the same kernel with a run-time plan.
Prints the list in the form _build.py holds it."""
import os, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = open(os.path.join(root, "include", "csdr_hb_taps.h")).read()
maxbw = [eval(x.strip()) for x in re.search(r"csdr_hb_maxbw\[[^\]]*\]\s*=\s*\{([^}]*)\}", h).group(1).split(",") if x.strip()]
lens = [int(x) for x in re.search(r"csdr_hb_len\[[^\]]*\]\s*=\s*\{([^}]*)\}", h).group(1).split(",") if x.strip()]
CIC3 = .5 - .4985

def plan(rate, bw):
    f, k = rate, []
    while bw > 0 and f > bw / maxbw[-1] and f > 7900.0 * 2.0 and len(k) < 9:
        if f >= bw / CIC3: k.append(3)
        else: k.append(next(lens[i] for i, m in enumerate(maxbw) if f >= bw / m))
        f /= 2.0
    return tuple(k)

RATES = [66666666.6667 / d for d in (1200.0, 600.0, 420.0, 340.0)] + [80.0e6 / d for d in (1280.0, 320.0, 128.0, 130.0, 40.0)]
MORE_RATES = [1.024e6, 2.048e6, 2.4e6, 2.5e6, 3.2e6, 8e6, 10e6]          # other common front ends (--more)
BWS = [1000.0, 10000.0, 15000.0, 20000.0]

def table(rates, skip=()):
    plans = {}
    for r in rates:
        for b in BWS:
            p = plan(r, b)
            if p and p not in skip: plans.setdefault(p, []).append((r, b))
    return plans

plans = table(RATES)
if __name__ == "__main__":
    import sys
    show = table(MORE_RATES, skip=plans) if "--more" in sys.argv else plans
    for p in sorted(show, key=lambda p: (len(p), p)):
        print("    %-44s # %s" % (str(p) + ",", ", ".join("%.0f/%.0f" % rb for rb in show[p])))
