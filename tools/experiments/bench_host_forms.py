#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer forms (what the reference's Qt host would see through the
drop-in classes): complex doubles in host memory in, results in host memory out, one channel.
Secondary measurement for DESIGN.md section 6 -- never the bench.py value."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import numpy as np
import cutesdr_amd as ca

out = {}
rng = np.random.default_rng(0)
n = 1 << 22
x = 3276.7 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
ff = ca.CFastFIR(16384); ff.SetupParameters(-5000, 5000, 0, 62500.0)
ff.ProcessData(x[:1 << 16]); ff.ProcessData(x)          # the first full-size call sizes the staging buffers
t0 = time.perf_counter(); reps = 5
for _ in range(reps): ff.ProcessData(x)
out["fastfir16384_host_MSps"] = round(reps * n / (time.perf_counter() - t0) / 1e6, 1)

d = ca.CDemodulator(2048); d.SetInputSampleRate(2e6); d.SetDemod(ca.DEMOD_FM, ca.fm_defaults()); d.SetDemodFreq(-100e3)
d.process_append(x[:1 << 16]); d.process_append(x)
t0 = time.perf_counter()
for _ in range(reps): d.process_append(x)
out["demod_chain_host_input_MSps"] = round(reps * n / (time.perf_counter() - t0) / 1e6, 1)

# raw 24-bit datagrams -> device unpack (3 B per component over PCIe instead of 8)
npk = 4096
raw = rng.integers(0, 256, (1, npk, 1444), dtype=np.uint8)
ca.unpack_packets_batch(raw, 1444)
t0 = time.perf_counter()
for _ in range(reps): ca.unpack_packets_batch(raw, 1444)
out["unpack24_host_MSps"] = round(reps * npk * 240 / (time.perf_counter() - t0) / 1e6, 1)
print(json.dumps(out))
