"""round 6: what does the GPU chain hand its SAM demodulator while the AGC's delay line still holds zeros (the first 468
samples of a stream)?  The oracle's PLL stays at rest on exact +0; on anything negative, however tiny, it is kicked by
atan2 = +-pi and the first two bursts differ by 0.48 / 1.70 of full scale (tools: /tmp exploration recorded in HISTORY)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cutesdr_amd as ca
from oracle import oracle
import test_postchain_gpu as T
from test_chain_parity_gpu import chain_input, pair
for nwin in (24, 40):
    d, r = pair(ca, oracle, "SAM")
    d.enable_taps(15); r.enable_taps(True)
    lim = d.buf_limit()
    x = chain_input("SAM", lim * nwin, 2e6)
    g3 = []; w3 = []; g2 = []; w2 = []
    for i in range(0, lim * 8, lim):
        r.clear_taps()
        kg, og = d.ProcessData(x[i:i + lim], True)
        kr, orr = r.ProcessData(x[i:i + lim], True)
        g3.append(d.tap(3)); w3.append(r.tap(3)); g2.append(d.tap(2)); w2.append(r.tap(2)); d.tap(1); d.tap(4)
    g3, w3, g2, w2 = map(np.concatenate, (g3, w3, g2, w2))
    nzg = np.nonzero(g3)[0]; nzw = np.nonzero(w3)[0]
    print(nwin, "first nonzero AGC output: gpu", nzg[:3], "oracle", nzw[:3], "len", len(g3))
    print("   gpu tap3[0:6]", g3[:6], "signbits re", np.signbit(g3.real[:468]).sum(), "im", np.signbit(g3.imag[:468]).sum())
    print("   gpu tap3[464:472]", g3[464:472]); print("   orc tap3[464:472]", w3[464:472])
    print("   gpu tap2[0:4]", g2[:4], " orc tap2[0:4]", w2[:4])
