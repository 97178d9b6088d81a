"""round 5: the headline kernel's energy per block -- socket power (rocm-smi) sampled while the launch repeats for a few
seconds, on noise and on zeros; joules per 16384-point block = average power x launch time / (256 channels x 64 blocks)."""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
C, T, n = 256, 1 << 19, 16384
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream

def power_w():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        for card in d.values():
            for k, v in card.items():
                if "ower" in k:
                    m = re.search(r"[0-9.]+", str(v))
                    if m: return float(m.group(0))
    except Exception as e:
        return None
    return None

def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        return [l.strip() for l in out.splitlines() if "sclk" in l or "mclk" in l][:3]
    except Exception:
        return None

def run(x, label, secs=6.0):
    y = torch.empty_like(x)
    ff = ca.FastFirBatch(C, n); ff.setup(-5000, 5000, 0, 62500.0)
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            p = power_w()
            if p is not None: samples.append(p)
            time.sleep(0.2)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); launches = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(200): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
        launches += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    clk = clocks()
    stop[0] = True; th.join()
    ms = e0.elapsed_time(e1) / launches
    tail = samples[len(samples) // 3:]                    # (the first third: ramp)
    p = sum(tail) / len(tail) if tail else None
    res = {"input": label, "ms_per_launch": round(ms, 4), "power_W_mean": None if p is None else round(p, 1), "power_samples": len(tail),
           "uJ_per_block": None if p is None else round(p * ms * 1e-3 / (C * (T // (n // 2))) * 1e6, 2), "clocks": clk}
    print(json.dumps(res), flush=True)
idle = power_w()
print(json.dumps({"idle_power_W": idle}))
noise = torch.randn((C, T, 2), device=dev) * 3276.7
run(noise, "noise -20 dBFS")
run(torch.zeros_like(noise), "zeros")
run(noise, "noise -20 dBFS (again)")
