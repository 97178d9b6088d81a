#!/usr/bin/env python3
"""Secondary measurements (not the bench.py contract): the other kernels of the path on BASELINE
config C4's per-GPU share -- 256 channels x 2^21 samples at 2 MSPS.
  K2  downconvert alone (FM chain 11,11,15,19,31 -> 62.5 kS/s): input MS/s, algorithmic GB/s
  chain  full CDemodulator batch (mixed AM/FM/USB), FastFIR 2048: input MS/s
Prints one JSON line."""
import json, sys, time, os
# the chain overlaps launches on several streams; HIP's default of 4 hardware queues serialises some of them
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 1 << 21
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
# synthetic receivers: every channel is tuned to its own carrier, -20 dBFS, AWGN -70 dBFS per
# component; channel c % 3: AM 50 % / 1 kHz, FM +-3 kHz / 1 kHz, two-tone SSB.  NOISE_ONLY=1: -20 dBFS noise
# (no carrier: the PLLs never lock and take their sample-by-sample path)
FS, A = 2e6, 3276.7
if os.environ.get("NOISE_ONLY"):
    x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * A
else:
    x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * (32767.0 * 10 ** (-70 / 20))
    t = torch.arange(T, device=dev, dtype=torch.float64) / FS
    for c in range(C):
        fc = 100e3 + 500.0 * c
        if c % 3 == 0:
            ph = 2 * torch.pi * fc * t; amp = A * (1.0 + 0.5 * torch.sin(2 * torch.pi * 1000.0 * t))
            x[c, :, 0] += (amp * torch.cos(ph)).float(); x[c, :, 1] += (amp * torch.sin(ph)).float()
        elif c % 3 == 1:
            ph = 2 * torch.pi * fc * t + 3.0 * torch.sin(2 * torch.pi * 1000.0 * t)
            x[c, :, 0] += (A * torch.cos(ph)).float(); x[c, :, 1] += (A * torch.sin(ph)).float()
        else:
            for off in (1200.0, 2340.0):
                ph = 2 * torch.pi * (fc + off) * t
                x[c, :, 0] += (0.5 * A * torch.cos(ph)).float(); x[c, :, 1] += (0.5 * A * torch.sin(ph)).float()
    del t, ph
y = torch.empty((C, T // 16, 2), device=dev, dtype=torch.float32)
stream = torch.cuda.current_stream().cuda_stream
out = {}

dc = ca.DownConvertBatch(C)
dc.set_data_rate(2e6, 15000.0)
for c in range(C):
    dc.set_frequency(-100e3 - 500.0 * c, channel=c)
def k2():
    dc.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T // 16, stream)
for _ in range(30): k2()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): k2()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
out["k2_ms"] = round(ms, 3)
out["k2_input_MSps"] = round(C * T / ms / 1e3, 1)
out["k2_alg_GBps"] = round(C * T * (8 + 8 / 32) / ms / 1e6, 1)

# K3: display spectrum, 4096-point frames (BASELINE config 1's transform), 256 channels x 512 frames, ave 4
fb = ca.FftBatch(C)
fb.set_params(4096, False, 0.0, 2e6); fb.set_ave(4)
def k3():
    fb.put_display_ptr(x.data_ptr(), T, 512, stream)
for _ in range(20): k3()
torch.cuda.synchronize()
e0.record()
for _ in range(30): k3()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
out["k3_ms"] = round(ms, 3)
out["k3_MSps"] = round(C * 512 * 4096 / ms / 1e3, 1)
out["k3_alg_GBps"] = round(C * 512 * 4096 * 12 / ms / 1e6, 1)

# K6: noise blanker at input rate (16 B per sample: 8 in + 8 out), wire-format unpack (6 B in + 8 B out per sample)
nbk = ca.NoiseProcBatch(C); nbk.setup(True, 50.0, 2.0, 2e6)
xb = torch.empty_like(x)
def k6():
    nbk.process_ptr(x.data_ptr(), T, T, xb.data_ptr(), T, stream)
for _ in range(15): k6()
torch.cuda.synchronize()
e0.record()
for _ in range(30): k6()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
out["k6_blanker_ms"] = round(ms, 3)
out["k6_blanker_alg_GBps"] = round(C * T * 16 / ms / 1e6, 1)
npk = T // 240
pk = torch.randint(0, 256, (C, npk, 1444), device=dev, dtype=torch.uint8)
def k6u():
    rc = ca.lib().csdr_ingest_unpack(0, pk.data_ptr(), C, npk, 1444, xb.data_ptr(), T, None, stream)
    assert rc == npk * 240
for _ in range(15): k6u()
torch.cuda.synchronize()
e0.record()
for _ in range(30): k6u()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
out["k6_unpack24_ms"] = round(ms, 3)
out["k6_unpack24_alg_GBps"] = round(C * npk * 240 * 14 / ms / 1e6, 1)
del xb, pk

if os.environ.get('K2_ONLY'):
    print(json.dumps(out)); sys.exit(0)

def info(**kw):
    base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
    base.update(kw); return ca.DemodInfo(**base)
modes = [(0, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)), (2, dict()),
         (3, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0))]
b = ca.DemodBatch(C, 2048)
b.set_input_rate(2e6)
for c in range(C):
    m, kw = modes[c % 3]
    b.set_demod(c, m, info(**kw))
b.commit()
for c in range(C):
    b.set_freq(c, -100e3 - 500.0 * c)
aud = torch.empty((C, T // 16 + 4096), device=dev, dtype=torch.float32)
def chain():
    b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), T // 16 + 4096, stream)
for _ in range(15): chain()
torch.cuda.synchronize()
e0.record()
for _ in range(30): chain()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
out["chain_ms"] = round(ms, 3)
out["chain_input_MSps"] = round(C * T / ms / 1e3, 1)
out["chain_alg_GBps"] = round(C * T * (8 + 4 / 32) / ms / 1e6, 1)
# the same chain fed with 24-bit datagrams (6 B per sample instead of 8): the down-converter, or the blanker in front
# of it, decodes them in its loads -- no unpack pass
npk = (T // 240) // 8 * 8          # 240 * 8 = 1920 = 64 * 30: a multiple of the largest decimation
Tp = npk * 240
# the receivers' own signals (x above) in the 24-bit wire format: value * 256, three little-endian bytes
pk = torch.zeros((C, npk, 1444), device=dev, dtype=torch.uint8)
for c0 in range(0, C, 32):
    v = torch.round(x[c0:c0 + 32, :Tp].reshape(-1, npk, 480) * 256.0).clamp(-(1 << 23), (1 << 23) - 1).to(torch.int32)
    body = torch.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], dim=-1).to(torch.uint8).reshape(-1, npk, 1440)
    pk[c0:c0 + 32, :, 4:] = body
    del v, body
nbk = ca.NoiseProcBatch(C); nbk.setup(True, 50.0, 2.0, 2e6)
def chain_pk(nb):
    rc = ca.lib().csdr_demod_batch_process_packets(b.h, pk.data_ptr(), npk, 1444, nb.h if nb is not None else None,
                                                   aud.data_ptr(), T // 16 + 4096, stream)
    assert rc == 0, ca._capi.last_error()
for name, nb in (("packets_chain_ms", None), ("packets_blanker_chain_ms", nbk)):
    for _ in range(10): chain_pk(nb)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30): chain_pk(nb)
    e1.record(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 30, 3)
# the same two with successive calls pipelined (csdr_demod_batch_set_pipelined): the blanker of call k+1 runs beside the
# post-chain of call k
# (only on request: the per-kernel averages of a profiled run should stay those of the strict chain)
if os.environ.get("CSDR_BENCH_CHAIN_PIPE"):
    b.set_pipelined(True)
for name, nb in ((("packets_chain_pipelined_ms", None), ("packets_blanker_chain_pipelined_ms", nbk)) if os.environ.get("CSDR_BENCH_CHAIN_PIPE") else ()):
    for _ in range(10): chain_pk(nb)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30): chain_pk(nb)
    e1.record(); b.flush(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 30, 3)
out["packets_samples_per_channel"] = Tp
out["channels"] = C
out["hw_queues"] = os.environ["GPU_MAX_HW_QUEUES"]
print(json.dumps(out))
