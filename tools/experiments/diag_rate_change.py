"""diagnostic (round 5): per-burst errors across input-rate changes, batch vs oracle and batch vs drop-in objects"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import cutesdr_amd as ca
from oracle import oracle
from test_postchain_gpu import info, burst_errors
from test_chain_parity_gpu import MODES, chain_input
from util_signals import FULL_SCALE
np.set_printoptions(linewidth=200, precision=2)

def streams(names, n, fs):
    t = np.arange(n)
    return np.stack([chain_input(m, n, fs) * np.exp(2j * np.pi * 700.0 * c * t / fs) for c, m in enumerate(names)]).astype(np.complex64)

def scenario(names, plan, chunked):
    C = len(names)
    b = ca.DemodBatch(C, 2048)
    b.set_input_rate(2e6)
    singles, refs = [], []
    for c, name in enumerate(names):
        m, kw = MODES[name]
        b.set_demod(c, m, info(ca, **kw))
        for lst, mod in ((singles, ca), (refs, oracle)):
            o = mod.CDemodulator(2048); o.SetInputSampleRate(2e6); o.SetDemod(m, info(mod, **kw)); o.SetDemodFreq(-100e3 - 700.0 * c)
            lst.append(o)
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3 - 700.0 * c)
    for step in plan:
        if step[0] == "rate":
            b.set_input_rate(step[1])
            for o in singles + refs: o.SetInputSampleRate(step[1])
            print("  -> rate", step[1], "groups", b.group_count(), [b.output_rate(c) for c in range(C)])
        elif step[0] == "demod":
            for c, name in enumerate(names):
                m, kw = MODES[name]
                b.set_demod(c, m, info(ca, **kw)); singles[c].SetDemod(m, info(ca, **kw)); refs[c].SetDemod(m, info(oracle, **kw))
            print("  -> SetDemod all")
        elif step[0] == "mode":
            c, name = step[1], step[2]
            m, kw = MODES[name]
            b.set_demod(c, m, info(ca, **kw)); singles[c].SetDemod(m, info(ca, **kw)); refs[c].SetDemod(m, info(oracle, **kw))
            names = list(names); names[c] = name
            print("  -> receiver", c, "now", name, "groups", b.group_count())
        else:
            n, fs = step[1], step[2]
            x = streams(names, n, fs)
            if chunked:
                got = [np.concatenate(p) for p in zip(*[b.process(x[:, i:i + chunked]) for i in range(0, n, chunked)])]
            else:
                got = b.process(x)
            for c, name in enumerate(names):
                one = singles[c].process_append(x[c].astype(np.complex128))
                want = refs[c].process_append(x[c].astype(np.complex128))
                k = min(len(got[c]), len(want)) // 1024 * 1024
                e = burst_errors(got[c][:k].astype(np.float64), want[:k]) / FULL_SCALE if k else np.array([])
                e1 = burst_errors(one[:k], want[:k]) / FULL_SCALE if k else np.array([])
                d = np.abs(got[c][:k] - one[:k]).max() / FULL_SCALE if k else 0
                print("   %-4s c=%d n=(%d,%d,%d) batch-vs-single %.2e | batch-vs-oracle %s | single-vs-oracle %s" % (
                    name, c, len(got[c]), len(one), len(want), d, np.array2string(e[:8], formatter={'float': lambda v: "%.1e" % v}),
                    np.array2string(e1[:8], formatter={'float': lambda v: "%.1e" % v})))

lim = 19968
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "split"):
    print("== split scenario")
    scenario(["USB", "USB", "USB", "AM", "AM", "FM"],
             [("run", 16 * lim, 2e6), ("mode", 1, "FM"), ("run", 16 * lim, 2e6), ("rate", 3.2e6), ("run", 16 * 32768, 3.2e6),
              ("rate", 2e6), ("run", 16 * lim, 2e6), ("run", 16 * lim, 2e6)], 0)
if which in ("all", "plain"):
    print("== plain scenario, one call")
    scenario(["FM", "AM", "USB", "CWU"], [("run", 16 * lim, 2e6), ("rate", 3.2e6), ("run", 16 * 32768, 3.2e6), ("rate", 2e6), ("run", 16 * lim, 2e6)], 0)
    print("== plain scenario, window by window")
    scenario(["FM", "AM", "USB", "CWU"], [("run", 16 * lim, 2e6), ("rate", 500e3), ("run", 16 * lim, 500e3), ("rate", 2e6), ("demod",), ("run", 16 * lim, 2e6)], lim)
