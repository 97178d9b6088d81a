"""round 5: are the down-converter's output WORDS independent of how a stream is cut into calls?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import cutesdr_amd as ca
from util_signals import fm_carrier
lim = 19968
x = fm_carrier(24 * lim, 2e6, 100e3).astype(np.complex64)[None, :]
def run(cuts):
    d = ca.DownConvertBatch(1)
    d.set_data_rate(2e6, 15000.0); d.set_frequency(-100e3)
    out, at = [], 0
    for c in cuts:
        out.append(d.process(x[:, at:at + c])[0]); at += c
    return np.concatenate(out)
a = run([24 * lim]); b = run([lim] * 24); c = run([8 * lim] * 3)
for name, y in (("24 windows", b), ("3 x 8 windows", c)):
    d = np.abs(a - y)
    print(name, "bit-equal" if np.array_equal(a.view(np.uint32), y.view(np.uint32)) else "max |diff| %.3g of full scale, %d of %d words differ" % (d.max() / 32767.0, int((a.view(np.uint32) != y.view(np.uint32)).sum()), a.size * 2))
