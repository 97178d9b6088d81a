"""round 6: why did SAM stereo differ from the oracle by 0.5 / 1.7 of full scale in its first two bursts in
tests/test_chain_taps_gpu.py while test_chain_every_mode_from_sample_zero[SAM-stereo] passes?  Cases: lone object; a second
object alive; a second object also processing; taps on."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cutesdr_amd as ca
from oracle import oracle
import test_postchain_gpu as T
from test_chain_parity_gpu import chain_input, pair
FS = T.FULL_SCALE
def errs_of(nwin, second, second_runs, taps, mode="SAM", stereo=True):
    d, r = pair(ca, oracle, mode)
    other = pair(ca, oracle, mode)[0] if second else None
    if taps: d.enable_taps(taps)
    lim = d.buf_limit()
    x = chain_input(mode, lim * nwin, 2e6)
    e = []
    for i in range(0, len(x), lim):
        kg, og = d.ProcessData(x[i:i + lim], stereo)
        kr, orr = r.ProcessData(x[i:i + lim], stereo)
        if second_runs: other.ProcessData(x[i:i + lim], stereo)
        assert kg == kr
        for j in range(0, kr, 1024): e.append(np.abs(og[j:j + 1024] - orr[j:j + 1024]).max() / FS)
    return np.array(e[:5])
for nwin in (24, 40):
    for second, runs, taps in ((False, False, 0), (True, False, 0), (True, True, 0), (False, False, 15), (False, False, 4), (False, False, 3)):
        print(nwin, "second" if second else "lone", "runs" if runs else "", "taps", taps, np.array2string(errs_of(nwin, second, runs, taps), precision=2), flush=True)
print("mono", np.array2string(errs_of(24, True, True, 15, stereo=False), precision=2))
