#!/usr/bin/env python3
"""The decimator plans (stage sequences of CDownConvert::SetDataRate, dsp/downconvert.cpp:127-166) behind the
reference's radios: its sample-rate tables (interface/sdrinterface.cpp:75-114: SDR-IQ, NetSDR, SDR-IP) x the
demodulators' maximum bandwidths (gui/mainwindow.cpp:1006-1050: AM/SAM 10 kHz, FM 15 kHz, SSB 20 kHz, CW 1 kHz),
and with --more the same bandwidths at other common front-end rates.  The down-converter is compiled for these by default
(cutesdr_amd/_build.py: default_dc_plans) and for EVERY sequence the selection rule can produce (all_dc_plans, 164
of them) with CSDR_ALL_DC_PLANS=1; any other plan runs with the plan taken at run time.  --all lists the whole set with a (rate, bandwidth) pair that selects each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutesdr_amd import _build

RATES, MORE_RATES, BWS = _build.RADIO_RATES, _build.MORE_RATES, _build.DEMOD_BWS
plan = _build.dc_plan

def table(rates, skip=()):
    tables = _build._hb_tables()
    plans = {}
    for r in rates:
        for b in BWS:
            p = plan(r, b, tables)
            if p and p not in skip: plans.setdefault(p, []).append((r, b))
    return plans

if __name__ == "__main__":
    if "--all" in sys.argv: show = {p: [rb] for p, rb in _build.all_dc_plans().items()}
    elif "--more" in sys.argv: show = table(MORE_RATES, skip=table(RATES))
    else: show = table(RATES)
    for p in sorted(show, key=lambda p: (len(p), p)):
        print("    %-44s # %s" % (str(p) + ",", ", ".join("%.0f/%.0f" % rb for rb in show[p])))
