import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import cutesdr_amd as ca
from oracle import oracle
import test_postchain_gpu as T
fs, C = 2e6, 1
names = ["FM"]
b = ca.DemodBatch(C, 2048); b.set_input_rate(fs)
m, kw = T.MODES["FM"]
b.set_demod(0, m, T.info(ca, **kw))
r = oracle.CDemodulator(2048); r.SetInputSampleRate(fs); r.SetDemod(m, T.info(oracle, **kw)); r.SetDemodFreq(-100e3)
b.commit(); b.set_freq(0, -100e3)
n = 19968 * 64
x = T.make_input("FM", 2 * n, fs)[None, :]
for part in (x[:, :n], x[:, n:]):
    got = b.process(part)[0]; want = r.process_append(part[0])
    d = np.abs(got - want).reshape(-1, 1024).max(axis=1)
    print(len(got), "bad bursts:", np.nonzero(d > 30)[0].tolist()[:20], "max", d.max())
    gz = (np.abs(got).reshape(-1, 1024).max(axis=1) == 0); wz = (np.abs(want).reshape(-1, 1024).max(axis=1) == 0)
    print(" squelched gpu:", np.nonzero(gz)[0].tolist()[:10], "oracle:", np.nonzero(wz)[0].tolist()[:10])
