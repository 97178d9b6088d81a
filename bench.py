#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X CuteSDR dsp/ receive chain.

Workload c3 (default, the config BASELINE.json's metric is quoted on): 256 batched channels x 2^19 complex
samples through the 16384-pt CFastFIR overlap-save on one MI355X, inputs and outputs resident in HBM.  One
"step" = one pass of all channels through the filter.  The same run also reports, as secondary objects of the
one JSON line: the distinct-filter variant of C3 (every channel its own H: +16 B/sample of filter traffic), a
post-timing parity spot check of the buffer that was just timed against the oracle, and the other BASELINE
configurations, each with the oracle's fp64 CPU path timed beside it on one host core (rank 0, N = 1, a bounded
sample): `chain_c4` (the per-GPU share of C4: 256 mixed AM/FM/USB receivers x 2^21 raw samples through the whole
CDemodulator chain, pipelined AND strict mode), `spectrum_c1` (4096-pt CFft display spectrum), `chain_c2` (one
receiver through the 16384-pt filter and the FM chain), `chain_c5` (one 10 MSPS receiver to 48 kHz audio), and the two
input-rate kernels alone: `downconv_k2`, `blanker_k6`.
Workload c4 makes the chain the primary metric instead.

Multi-GPU: channels are independent, so every rank owns its own channels (contiguous channel ranges, weak
scaling) and the data path has no collective; torch.distributed carries the barrier, the MAX-reduce of the
elapsed time and -- for the chain -- the optional gather of per-channel S-meter + audio to rank 0 (RCCL,
timed separately, never inside `value`).  `--gpus N` without a launcher starts N rank processes itself
(fresh children, spawned before this process touches torch or HIP); under torch.distributed.run the
RANK/LOCAL_RANK/WORLD_SIZE environment is used as it is.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FFT_N = 16384
CHANNELS = 256
T_PER_CH = 1 << 19
FS = 62500.0
ALG_BYTES_PER_SAMPLE = 16.0       # 8 B fp32 I/Q read + 8 B written
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "complex IQ MSamples/s through CFastFIR+demod chain; achieved HBM GB/s vs peak"
PREWARM_S = 0.25                  # untimed, before the W warm-up steps: the first launches of a process run ~13 % slow
K1_SOURCES = ("cutesdr_amd/csrc/fastfir2_kernels.hip", "cutesdr_amd/csrc/fastfir_kernels.hip",
              "cutesdr_amd/csrc/fastfir_dev.hpp", "cutesdr_amd/csrc/fft_core.hpp", "cutesdr_amd/csrc/capi_fastfir.hip")
C4_T = 1 << 21                    # raw samples per receiver and step of the chain workload
C4_FS = 2e6


# ---------------------------------------------------------------- distributed plumbing
def dist_init(backend=None):
    """One process per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment."""
    import torch.distributed as dist
    ctx = SimpleNamespace(rank=int(os.environ.get("RANK", "0")), world=int(os.environ.get("WORLD_SIZE", "1")),
                          local=int(os.environ.get("LOCAL_RANK", "0")), dist=dist, backend=backend)
    if os.environ.get("CSDR_BENCH_ONE_GPU"):
        # rehearsal of the multi-rank code on a one-GPU box (tests): every rank on cuda:0, collectives over gloo
        ctx.local, backend = 0, "gloo"
        ctx.backend = backend
    # CSDR_BENCH_FORCE_DIST=1: a process group even for ONE rank, so that the barrier, the MAX-reduce, the census and the
    # chain's gather go through the real library (RCCL) on a one-GPU box -- the world == 1 short-cuts are switched off
    # (tests/test_bench_dist.py::test_one_rank_through_rccl)
    ctx.dist_on = ctx.world > 1 or bool(os.environ.get("CSDR_BENCH_FORCE_DIST"))
    if ctx.dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ctx.world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend is None:
            import torch
            ndev = torch.cuda.device_count()
            if ctx.local >= ndev:                     # fail before RCCL does, with a message that says why
                raise SystemExit("bench.py: rank %d wants cuda:%d but this node shows %d device(s) -- run with "
                                 "--gpus <= %d" % (ctx.rank, ctx.local, ndev, ndev))
            ctx.backend = "nccl"                      # = RCCL on ROCm
            dist.init_process_group("nccl", device_id=torch.device("cuda", ctx.local))
        else:
            dist.init_process_group(backend)
    return ctx


def rank_census(ctx):
    """What the process group actually is, for the JSON line: backend, world size as the group reports it, and every
    rank's device (index, name, PCI bus id, UUID where the runtime has one) gathered to rank 0 -- the driver can check
    that RCCL saw N ranks on N different devices."""
    import torch
    me = {"rank": ctx.rank, "local_rank": ctx.local}
    if torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(ctx.local)
        me.update(device=ctx.local, name=pr.name, pci_bus_id=getattr(pr, "pci_bus_id", None),
                  uuid=str(getattr(pr, "uuid", "")) or None)
    if not getattr(ctx, "dist_on", ctx.world > 1):
        return {"backend": None, "world_size": 1, "ranks": [me]}
    everyone = [None] * ctx.world
    ctx.dist.all_gather_object(everyone, me)
    return {"backend": ctx.dist.get_backend(), "world_size": ctx.dist.get_world_size(), "ranks": everyone}


def shard_channels(ctx, total_channels):
    """Contiguous channel range of this rank: [lo, hi)."""
    per = total_channels // ctx.world
    return ctx.rank * per, (ctx.rank + 1) * per


def dist_barrier(ctx):
    if getattr(ctx, "dist_on", ctx.world > 1):
        ctx.dist.barrier()


def dist_max(ctx, value):
    if not getattr(ctx, "dist_on", ctx.world > 1):
        return float(value)
    import torch
    dev = torch.device("cuda", ctx.local) if ctx.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    ctx.dist.all_reduce(t, op=ctx.dist.ReduceOp.MAX)
    return float(t.item())


def dist_finish(ctx):
    if getattr(ctx, "dist_on", ctx.world > 1):
        ctx.dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """Start n rank processes of this script (one GPU each), relay rank 0's output, return the worst exit code.
    Runs before anything in this process has imported torch or touched HIP: the children are fresh
    interpreters, nothing is re-executed in place."""
    if "--stub" not in argv and not os.environ.get("CSDR_BENCH_ONE_GPU"):
        import torch                                  # (device_count() may initialise HSA in this parent when amdsmi is not
        ndev = torch.cuda.device_count()              # there; harmless: the ranks are fresh children, nothing is re-exec'd)
        if ndev < n:
            print("bench.py: --gpus %d but this node shows %d device(s); nothing started" % (n, ndev), file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        # RCCL between processes needs dmabuf IPC on this driver (HSA_ENABLE_IPC_MODE_LEGACY=0): the image exports it;
        # setdefault never overrides what the environment says, and covers a host that does not export it
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else subprocess.DEVNULL          # rank 0 prints the one JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    deadline = time.time() + 3000
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is not None:
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:                          # a rank died: the others would wait for it forever
                        q.terminate()
        if time.time() > deadline:
            for q in pending:
                q.kill()
            return 124
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------- result line
def roofline_obj(achieved_gbs, kern_ms, kernel, alg_bytes, traffic):
    return {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel,
            "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}


def result_line(ctx, channels, samples, steps, warmup, elapsed, kern_ms, traffic=None, cpu=None, workload="c3",
                extra=None):
    per_step = channels * samples                     # samples one rank processes per step
    value = per_step * steps * ctx.world / elapsed / 1e6
    if workload == "c3":
        alg = ALG_BYTES_PER_SAMPLE * per_step
        wl = ("C3: %d channels/GPU x 2^%d IQ samples, 16384-pt CFastFIR overlap-save (8193 taps, hop 8192), "
              "shared -5..+5 kHz filter @62.5 kS/s" % (channels, samples.bit_length() - 1))
        kernel = "csdr::fastfir_os2_kernel<14>"
        cfg = {"workload": wl, "channels_per_gpu": channels, "samples_per_channel": samples, "fft_size": FFT_N}
    else:
        alg = (8.0 + 4.0 / 32.0) * per_step           # 8 B raw I/Q in + 4 B mono audio out per 32 raw samples
        wl = ("C4 per-GPU share: %d mixed AM/FM/USB receivers/GPU x 2^%d raw IQ samples @2 MSPS through the whole "
              "CDemodulator chain (downconvert, 2048-pt CFastFIR, S-meter, AGC, demodulator)"
              % (channels, samples.bit_length() - 1))
        kernel = "whole chain: every launch of csdr_demod_batch_process (downconv_kernel dominant)"
        cfg = {"workload": wl, "channels_per_gpu": channels, "samples_per_channel": samples, "fft_size": 2048}
    cfg["parallelism"] = "channels sharded x%d, no data-path collective" % ctx.world
    cfg["prewarm_s"] = PREWARM_S
    line = {
        "metric": METRIC, "value": round(value, 2), "unit": "MSamples/s",
        "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "config": cfg,
        "roofline": roofline_obj(alg / (kern_ms * 1e-3) / 1e9, kern_ms, kernel, alg, traffic),
        "cpu_baseline": cpu,
    }
    line["roofline"]["frac_of_measured_copy"] = round(alg / (kern_ms * 1e-3) / 1e9 / COPY_MEASURED_GBS, 4)
    if extra:
        pc = extra.pop("power_clock", None)
        if pc:                                            # socket power / shader clock while the headline launch repeats
            line["roofline"].update({k: pc[k] for k in ("power_w", "sclk_ghz") if k in pc})
            line["roofline"]["power_source"] = pc.get("power_source")
        others, secondary = condensed(extra)
        if others:
            line["roofline"]["others"] = others           # scalars only: the driver keeps `roofline` and `config` whole
        if secondary:
            line["config"]["secondary"] = secondary
        line.update(extra)
    return line


COPY_MEASURED_GBS = 6300.0        # what a device-to-device copy reaches on this part: the practical ceiling


def condensed(extra):
    """The secondary measurements as SCALARS under the keys the driver's record keeps (`roofline`, `config`): every
    other top-level object of the line is preserved only as far as the tail of stdout reaches.  -> (others, secondary)"""
    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    def roof(obj, ms_key="ms_per_launch"):
        if not obj:
            return None
        r = {"ms": g(obj, ms_key), "frac": g(obj, "roofline", "frac"), "traffic_over_algorithmic": g(obj, "roofline", "traffic_over_algorithmic"),
             "parity_ok": g(obj, "parity_checked", "ok")}
        return {k: v for k, v in r.items() if v is not None}
    others = {}
    c4 = extra.get("chain_c4")
    if c4:
        others["chain_c4_strict"] = {k: v for k, v in {
            "ms": g(c4, "strict", "ms_per_step"), "event_ms": g(c4, "strict", "event_ms_per_step"), "frac": g(c4, "strict", "frac_of_hbm_peak"),
            "traffic_over_algorithmic": g(c4, "roofline", "traffic_over_algorithmic"), "parity_ok": g(c4, "parity_checked", "ok")}.items() if v is not None}
        others["chain_c4_pipelined"] = {"ms": g(c4, "ms_per_step"), "frac": g(c4, "frac_of_hbm_peak")}
    for key, name in (("downconv_k2", "downconv_k2"), ("spectrum_c1", "spectrum_c1"), ("blanker_k6", "blanker_k6")):
        if extra.get(key):
            others[name] = roof(extra[key])
    mk = g(extra, "packets_chain", "mask_kernel")
    if mk:
        others["mask_kernel"] = roof(mk)
    for size in ("2048", "4096", "8192"):
        o = g(extra, "fastfir_sizes", "fastfir", size)
        if o:
            others["fastfir_" + size] = {"ms": o.get("ms"), "frac": o.get("frac"), "parity_ok": g(o, "parity_checked", "ok")}
    for size in ("8192", "16384"):
        o = g(extra, "fastfir_sizes", "spectrum", size)
        if o:
            others["spectrum_" + size] = {"ms": o.get("ms"), "frac": o.get("frac"), "parity_ok": g(o, "parity_checked", "ok")}
    sec = {"host_form_MSps": g(extra, "host_form", "raw_input_MSamples_per_s"), "host_form_parity_ok": g(extra, "host_form", "parity_checked", "ok"),
           "host_form_deferred_MSps": g(extra, "host_form", "deferred_output", "raw_input_MSamples_per_s"),
           "host_form_deferred_same_words": g(extra, "host_form", "deferred_output", "same_words_one_window_later"),
           "chain_c2_ms": g(extra, "chain_c2", "ms_per_call"), "chain_c2_parity_ok": g(extra, "chain_c2", "parity_checked", "ok"),
           "chain_c5_ms": g(extra, "chain_c5", "ms_per_call"), "chain_c5_parity_ok": g(extra, "chain_c5", "parity_checked", "ok"),
           "packets_chain_ms": g(extra, "packets_chain", "packets_chain_ms"),
           "packets_blanker_chain_ms": g(extra, "packets_chain", "packets_blanker_chain_ms"),
           "packets16_blanker_chain_ms": g(extra, "packets_chain", "packets16_blanker_chain_ms"),
           "packets_parity_ok": g(extra, "packets_chain", "parity_checked", "ok"),
           "retune_us": g(extra, "control_plane", "retune_us"), "retune_step_increase_ms": g(extra, "control_plane", "step_increase_ms"),
           "retune_parity_ok": g(extra, "control_plane", "parity_ok")}
    return others, {k: v for k, v in sec.items() if v is not None}


def k1_source_hash():
    h = hashlib.sha256()
    for rel in K1_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profiled_traffic():
    """HBM bytes per launch of the headline kernel from the committed PMC profile -- only if that profile was
    taken on exactly these kernel sources (profiles/traffic_latest.json carries their hash); null otherwise."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        d = json.load(open(path))
        if d.get("k1_source_sha16") == k1_source_hash():
            return d.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def live_traffic(timeout_s=240):
    """HBM bytes per launch MEASURED IN THIS RUN for the headline kernel and for the secondary objects: two child runs
    of this script (`--pmc-child`: a few launches of every workload) under `rocprofv3 --pmc`, FETCH_SIZE and
    WRITE_SIZE in passes of their own as the guide prescribes (never with a trace domain), per-dispatch values grouped
    by kernel and combined with the guide's gfx950 correction (FETCH_SIZE counts half of a coalesced read; calibrated
    for 4-, 8- and 16-byte loads in profiles/r03_fetch_calib.json): (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Rank 0 at
    N = 1 only, after everything timed.  None when the profiler is not there or a pass fails -- the caller then falls
    back to the committed profile (same-sources hash) or null."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="csdr_pmc_")
    per = {}                                              # counter -> object -> KiB per launch / step
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child"]
            env = dict(os.environ, TMPDIR=out)
            r = subprocess.run(cmd, cwd=out, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            if r.returncode != 0:
                return None
            counts = None
            for line in r.stdout.decode(errors="replace").splitlines():
                if line.startswith('{"pmc_child"'):
                    counts = json.loads(line)["pmc_child"]
            if not counts:
                return None
            acc = {"k1": [], "k2": [], "k3": 0.0, "k6": [], "k6m": [], "chain": 0.0}
            by_name = {}                                  # the chain steps' share, kernel by kernel
            dcs = []                                      # every down-converter dispatch: (dispatch id, value)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]
                    if row["Counter_Name"] != counter or "csdr" not in name:
                        continue
                    v = float(row["Counter_Value"])
                    if "fastfir_os2_kernel<14>" in name:
                        acc["k1"].append(v)
                    elif "downconv_kernel" in name:
                        dcs.append((int(row["Dispatch_Id"]), v))
                    elif "spectrum" in name:
                        acc["k3"] += v
                    elif "noiseblank_kernel<true" in name or "noiseblank_mask_int_kernel" in name:
                        acc["k6m"].append(v)
                    elif "noiseblank_kernel" in name:
                        acc["k6"].append(v)
                    else:
                        acc["chain"] += v
                        short = name.split("csdr::", 1)[-1].split("<")[0].split("(")[0]
                        by_name[short] = by_name.get(short, 0.0) + v
            # the child runs "K2 alone" BEFORE its chain steps and says how many launches that was: the first counts["k2"]
            # down-converter dispatches are those, whatever grid the library's segment rule gave them (the grid size
            # depends on the channel count, the workgroup budget and CSDR_DC_WGS); the rest belong to the chain steps
            dcs.sort()
            acc["k2"] = [v for _, v in dcs[:counts["k2"]]]
            acc["chain"] += sum(v for _, v in dcs[counts["k2"]:])
            by_name["downconv_kernel"] = sum(v for _, v in dcs[counts["k2"]:])
            if len(acc["k1"]) != counts["k1"] or len(acc["k2"]) != counts["k2"] or not acc["k6"] or not acc["k6m"] or \
                    len(dcs) <= counts["k2"]:
                return None                               # a phase is missing or split differently: no figure rather than a wrong one
            mean = lambda v: sum(v) / len(v) if v else None
            per[counter] = {"k1": mean(acc["k1"]), "k2": mean(acc["k2"]), "k6": sum(acc["k6"]) / counts["k6"],
                            "k6m": sum(acc["k6m"]) / counts["k6m"],
                            "k3": acc["k3"] / counts["k3"] if counts.get("k3") else None,
                            "chain": acc["chain"] / counts["chain"] if counts.get("chain") else None,
                            "chain_by_kernel": {k: v / counts["chain"] for k, v in by_name.items()} if counts.get("chain") else None}
        res = {}
        for key in ("k1", "k2", "k3", "k6", "k6m", "chain"):
            f, w = per["FETCH_SIZE"].get(key), per["WRITE_SIZE"].get(key)
            res[key] = None if f is None or w is None else {"bytes": (2.0 * f + w) * 1024.0, "FETCH_SIZE_KiB": round(f, 1),
                                                            "WRITE_SIZE_KiB": round(w, 1)}
        fk, wk = per["FETCH_SIZE"].get("chain_by_kernel"), per["WRITE_SIZE"].get("chain_by_kernel")
        if fk and wk and res.get("chain"):
            res["chain"]["GB_by_kernel"] = {k: round((2.0 * fk.get(k, 0.0) + wk.get(k, 0.0)) * 1024.0 / 1e9, 3)
                                            for k in sorted(set(fk) | set(wk))}
        res["source"] = ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this script (KiB means per launch; "
                         "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 correction of MI355X_MICROARCH.md)")
        return res
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def run_pmc_child():
    """A few launches of every workload for the counter passes of live_traffic(): C3 (K1), then on the C4 buffer K2
    alone, K3, K6 and whole strict chain steps.  Prints how many launches / steps of each ran."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import cutesdr_amd as ca
    ctx = dist_init()
    torch.cuda.set_device(ctx.local)
    w = C3Workload(torch, ca, ctx, CHANNELS)
    for _ in range(6):
        w.step()
    torch.cuda.synchronize()
    del w
    torch.cuda.empty_cache()
    c4 = C4Workload(torch, ca, ctx, CHANNELS)
    st = torch.cuda.current_stream().cuda_stream
    C, T = c4.x.shape[0], c4.x.shape[1]
    dc = ca.DownConvertBatch(C, device=ctx.local)
    dc.set_data_rate(C4_FS, 15000.0)
    y = torch.empty((C, T // 16, 2), device=c4.x.device, dtype=torch.float32)
    for _ in range(3):
        dc.process_ptr(c4.x.data_ptr(), T, T, y.data_ptr(), T // 16, st)
    torch.cuda.synchronize()
    del dc, y
    fb = ca.FftBatch(C, device=ctx.local)
    fb.set_params(4096, False, 0.0, C4_FS); fb.set_ave(1)
    for _ in range(3):
        fb.put_display_ptr(c4.x.data_ptr(), T, 512, st)
    torch.cuda.synchronize()
    del fb
    nb = ca.NoiseProcBatch(C, device=ctx.local)
    nb.setup(True, 50.0, 2.0, C4_FS)
    xb = torch.empty_like(c4.x)
    for _ in range(3):
        nb.process_ptr(c4.x.data_ptr(), T, T, xb.data_ptr(), T, st)
    torch.cuda.synchronize()
    del nb, xb
    npk = (T // 240) // 8 * 8
    pk = datagrams_of(torch, c4.x, npk)
    nb = ca.NoiseProcBatch(C, device=ctx.local)
    nb.setup(True, 50.0, 2.0, C4_FS)
    run, mask = blank_mask_call(torch, ca, nb, pk, npk, st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    del nb, pk, mask
    c4.set_mode(False)
    for _ in range(12):                                   # (twelve: the first step of a fresh object touches its buffers for the first time)
        c4.step()
    torch.cuda.synchronize()
    print(json.dumps({"pmc_child": {"k1": 6, "k2": 3, "k3": 3, "k6": 3, "k6m": 3, "chain": 12}}), flush=True)


# ---------------------------------------------------------------- CPU baseline
def cpu_baseline(budget_s=18.0):
    """The fp64 CPU restatement (oracle, kind 'port') of the same filter on a bounded sample of the
    same workload: hops of one channel on 1 thread (the reference runs all DSP on one thread) until
    a third of budget_s has elapsed, with and without the reference's log10 side effect; then one
    channel per host core on all cores (channels are independent), faithful variant."""
    import numpy as np
    import threading
    from oracle import oracle as orc
    rng = np.random.default_rng(1)
    chunk = 1 << 20
    x = 3276.7 * (rng.standard_normal(chunk) + 1j * rng.standard_normal(chunk))
    res = {}
    budget_s = budget_s * 2.0 / 3.0
    ncores = min(os.cpu_count() or 1, 64)
    if hasattr(os, "sched_getaffinity"):
        ncores = min(ncores, len(os.sched_getaffinity(0)))
    counts = [0] * ncores

    filters = []
    for _ in range(ncores):
        ff = orc.CFastFIR(FFT_N)
        ff.set_faithful(1)
        ff.SetupParameters(-5000, 5000, 0, FS)
        filters.append(ff)
    small = x[:1 << 18]

    def worker(k, t_end):
        while time.perf_counter() < t_end:                # ctypes drops the GIL inside the C call
            filters[k].ProcessData(small)
            counts[k] += len(small)

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k, t0 + budget_s / 4)) for k in range(ncores)]
    for t in th: t.start()
    for t in th: t.join()
    res["allcores"] = sum(counts) / (time.perf_counter() - t0) / 1e6
    for name, faithful in (("faithful", 1), ("lean", 0)):
        ff = orc.CFastFIR(FFT_N)
        ff.set_faithful(faithful)
        ff.SetupParameters(-5000, 5000, 0, FS)
        ff.ProcessData(x[:FFT_N])                       # warm
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / 2:
            ff.ProcessData(x)
            n += chunk
        res[name] = n / (time.perf_counter() - t0) / 1e6
        res[name + "_samples"] = n
    return {
        "value": round(res["faithful"], 3), "unit": "MSamples/s", "cores": 1, "kind": "port",
        "sample": "1 channel of the C3 stream through the fp64 16384-pt oracle FastFIR: %d samples with the "
                  "reference's per-FFT power/log10 side effect (dsp/fft.cpp:564-589) kept = value, %d samples "
                  "without it = lean_value" % (res["faithful_samples"], res["lean_samples"]),
        "lean_value": round(res["lean"], 3),
        "allcores_value": round(res["allcores"], 3), "allcores": ncores,
        "true_reference_note": "the reference's own CFastFIR patched to 16384/8193 ran at 14.5 MSamples/s on one "
                               "Xeon 2.1 GHz thread in the survey container (BASELINE.md section 2); it cannot be "
                               "built on this box (Qt headers), so the timed code is the fp64 port",
    }


def cpu_rate(run_once, samples_per_call, budget_s):
    """run_once() on this thread until budget_s has elapsed (at least once): (MSamples/s, samples processed)"""
    n, t0 = 0, time.perf_counter()
    while True:
        run_once()
        n += samples_per_call
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            return n / dt / 1e6, n


def cpu_obj(value, samples, what):
    return {"value": round(value, 3), "unit": "MSamples/s", "cores": 1, "kind": "port",
            "sample": "%s: %d samples on one host core (fp64 oracle)" % (what, samples)}


def gpu_ms(torch, fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


FULL_SCALE = 32767.0


def chain_burst_check(got, want, mode, hop=1024, fm_late=0):
    """The chain rule the parity tests apply burst by burst (tests/test_postchain_gpu.py: check_chain_bursts; a burst =
    one FastFIR hop of audio: 1024 samples behind the 2048-point filter, 8192 behind the 16384-point one), on the common prefix of the product's audio and the oracle's.  FM: 1e-3 of
    full scale from the 4th burst, 3e-5 from the 7th (its first bursts demodulate the filter's start-up, DESIGN
    section 5); the other modes: 5e-4 from sample 0, 2e-5 from the 3rd burst."""
    import numpy as np
    n = (min(len(got), len(want)) // hop) * hop
    if n == 0:
        return {"bursts": 0, "ok": False}
    e = np.abs(np.asarray(got[:n], dtype=np.float64) - np.asarray(want[:n], dtype=np.float64)).reshape(-1, hop).max(axis=1) / FULL_SCALE
    idx = np.arange(len(e))
    if mode == "FM":
        # the derived start-up rule of the tests (tests/startup_bounds.py: the oracle's own per-burst spread under an fp32
        # filter's error floor, x 2): k bursts behind the burst in which the loop pulls in -- the stream's first burst with
        # audio; a 10 MSPS chain's first burst is silent on both sides -- 5.6e-2, 9.8e-3, 2.0e-3, 3.8e-4, 7.4e-5 of full scale,
        # then the steady bound.  `early` reports bursts 3 .. 5 behind the pull-in against the bound of the first of them.
        startup = [None, 5.6e-2, 9.8e-3, 2.0e-3, 3.8e-4, 7.4e-5]
        big = np.nonzero(e[:3] > 0.2)[0]
        start = (int(big[0]) if len(big) else 0) + fm_late
        ok = bool(np.isfinite(e).all() and (e <= 2.5).all())
        for k in range(1, len(startup)):
            ok = ok and bool((e[idx >= start + k] <= startup[k]).all())
        early, steady, t_early, t_steady = e[(idx >= start + 3) & (idx < start + 6)], e[idx >= start + 6], startup[3], 3e-5
        ok = ok and bool((steady <= t_steady).all())
    else:
        early, steady, t_early, t_steady = e[idx < 2], e[idx >= 2], 5e-4, 2e-5
        ok = bool((early <= t_early).all() and (steady <= t_steady).all())
    return {"bursts": int(len(e)), "max_err_early_over_full_scale": float(early.max()) if len(early) else None,
            "max_err_steady_over_full_scale": float(steady.max()) if len(steady) else None,
            "tolerance_early": t_early, "tolerance_steady": t_steady, "ok": ok}


def fm_defaults(mod):
    return mod.DemodInfo(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                         FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                         AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)


def fm_stream(torch, dev, n, fs, fc):
    """a -20 dBFS carrier at fc, FM +-3 kHz / 1 kHz, AWGN -70 dBFS per component: [1, n, 2] fp32 on the device"""
    t = torch.arange(n, device=dev, dtype=torch.float64) / fs
    ph = 2 * torch.pi * fc * t + 3.0 * torch.sin(2 * torch.pi * 1000.0 * t)
    x = torch.stack([(3276.7 * torch.cos(ph)).float(), (3276.7 * torch.sin(ph)).float()], dim=-1).reshape(1, n, 2).contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(0xF3 + int(fs) % 1000003)                # the same stream in every run: the check below is not a lottery
    x += torch.randn(x.shape, generator=g, device=dev, dtype=x.dtype) * (32767.0 * 10 ** (-70 / 20))
    return x


def to_c128(x_dev_row):
    import numpy as np
    a = x_dev_row.cpu().numpy().astype(np.float64)
    return a[:, 0] + 1j * a[:, 1]


def spectrum_c1(torch, ca, ctx, x, with_cpu, check=True):
    """BASELINE config C1's transform (dsp/fft.cpp:267-288: 4096 points, Hann, ave 1) on every channel of the
    resident buffer x [C, T, 2]: 512 frames per channel and launch.  SURVEY 8(d) prices a bin at 8 B in + 4 B out;
    the kernel keeps the running sums in registers and writes only the LAST frame's bels, so both fractions are
    given: at 12 B/bin and on the bytes it actually moves."""
    C, T, N, frames = x.shape[0], x.shape[1], 4096, 512
    fb = ca.FftBatch(C, device=ctx.local)
    fb.set_params(N, False, 0.0, C4_FS)
    fb.set_ave(1)
    st = torch.cuda.current_stream().cuda_stream
    ms = gpu_ms(torch, lambda: fb.put_display_ptr(x.data_ptr(), T, frames, st), 10, 20)
    bins = C * frames * N
    moved = bins * 8.0 + C * N * 4.0
    # headline of this object: the bytes the kernel MOVES (8 B per bin read + the last frame's bels written); the
    # survey's 12 B per bin (every frame's 4-byte result counted, two thirds of which the kernel never writes) is kept
    # beside it under its own name
    out = {"config": "C1: 4096-pt CFft display spectrum (Hann, ave 1) @2 MSPS, %d channels x %d frames per launch" % (C, frames),
           "kernel": "csdr::spectrum16_kernel (4096 points as 256 threads x 16; + its frame-group combine)", "ms_per_launch": round(ms, 4),
           "MSamples_per_s": round(bins / ms / 1e3, 1),
           "roofline": roofline_obj(moved / ms / 1e6, ms, "csdr::spectrum16_kernel + spectrum_alpha / combine / count kernels", moved, None),
           "frac_of_hbm_peak": round(moved / ms / 1e6 / HBM_PEAK_GBS, 4),
           "at_survey_12B_per_bin": {"algorithmic_GBps": round(bins * 12.0 / ms / 1e6, 1),
                                     "frac": round(bins * 12.0 / ms / 1e6 / HBM_PEAK_GBS, 4)},
           "cpu_baseline": None}
    if with_cpu:
        from oracle import oracle as orc
        f = orc.CFft()
        f.SetFFTParams(N, False, 0.0, C4_FS)
        f.SetFFTAve(1)
        xs = to_c128(x[0, :64 * N])
        def once():
            for i in range(0, len(xs), N):
                f.PutInDisplayFFT(xs[i:i + N])
        v, n = cpu_rate(once, len(xs), 2.0)
        out["cpu_baseline"] = cpu_obj(v, n, "CFft::PutInDisplayFFT, 4096-pt frames of channel 0 (BASELINE configs[0])")
    if check:                                             # no averaging: the display holds the call's LAST frame
        import numpy as np
        from oracle import oracle as orc
        torch.cuda.synchronize()
        c = C // 3
        r = orc.CFft(); r.SetFFTParams(N, False, 0.0, C4_FS); r.SetFFTAve(1)
        r.PutInDisplayFFT(to_c128(x[c, (frames - 1) * N:frames * N]))
        want = np.asarray(r.ave_buf())
        got = fb.ave_buf(c).astype(np.float64)
        near = want > want.max() - 6.0
        err = float(np.abs(got - want)[near].max())
        out["parity_checked"] = {"channel": c, "bins_within_60dB_of_peak": int(near.sum()), "max_err_bels": err, "tolerance_bels": 0.001,
                                 "ok": bool(err <= 0.001)}
    del fb
    return out


def input_rate_kernels(torch, ca, ctx, x, with_cpu, check=True):
    """The two input-rate kernels of the path alone on the resident buffer x [C, T, 2]: K2, the down-converter (one plan
    group of all C receivers: 2 MSPS, FM bandwidth -> chain 11,11,15,19,31, 62.5 kS/s; 8 B in + 8 B out / 32 per
    sample) and K6, the noise blanker (8 B in + 8 B out per sample), each with the oracle on one host core."""
    C, T = x.shape[0], x.shape[1]
    dev = x.device
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    dc = ca.DownConvertBatch(C, device=ctx.local)
    dc.set_data_rate(C4_FS, 15000.0)
    for c in range(C):
        dc.set_frequency(-100e3 - 500.0 * c, channel=c)
    y = torch.empty((C, T // 16, 2), device=dev, dtype=torch.float32)
    ms = gpu_ms(torch, lambda: dc.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T // 16, st), 10, 20)
    alg = C * T * (8.0 + 8.0 / 32.0)
    out["downconv_k2"] = {"config": "K2 alone: %d receivers x 2^%d samples @2 MSPS, chain 11,11,15,19,31 -> 62.5 kS/s" % (C, T.bit_length() - 1),
                          "kernel": "csdr::downconv_kernel<DcPlanT<11,11,15,19,31>>", "ms_per_launch": round(ms, 4),
                          "raw_input_MSamples_per_s": round(C * T / ms / 1e3, 1),
                          "roofline": roofline_obj(alg / ms / 1e6, ms, "csdr::downconv_kernel<DcPlanT<11,11,15,19,31>>", alg, None),
                          "frac_of_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "cpu_baseline": None}
    if check:                                             # a FRESH object, one call on the timed buffer, receiver C/3 vs the oracle
        import numpy as np
        from oracle import oracle as orc
        c = C // 3
        dc2 = ca.DownConvertBatch(C, device=ctx.local)
        dc2.set_data_rate(C4_FS, 15000.0)
        for k in range(C):
            dc2.set_frequency(-100e3 - 500.0 * k, channel=k)
        dc2.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T // 16, st)
        torch.cuda.synchronize()
        got = y[c, :T // 32].cpu().numpy().astype(np.float64)
        r = orc.CDownConvert(); r.SetDataRate(C4_FS, 15000.0); r.SetFrequency(-100e3 - 500.0 * c)
        xc = to_c128(x[c])                                # (the reference's call pattern: m_InBufLimit windows)
        want = np.concatenate([np.asarray(r.ProcessData(xc[i:i + 19968])) for i in range(0, len(xc), 19968)])
        m = min(len(want), T // 32)
        err = float(np.abs((got[:m, 0] + 1j * got[:m, 1]) - want[:m]).max() / 32767.0) if m else float("inf")
        out["downconv_k2"]["parity_checked"] = {"receiver": c, "samples": int(T), "outputs_compared": int(m),
                                                "max_err_over_full_scale": err, "tolerance": 1e-5,
                                                "ok": bool(m >= T // 32 - 1 and err <= 1e-5)}
        del dc2
    del y, dc
    nb = ca.NoiseProcBatch(C, device=ctx.local)
    nb.setup(True, 50.0, 2.0, C4_FS)
    xb = torch.empty_like(x)
    ms = gpu_ms(torch, lambda: nb.process_ptr(x.data_ptr(), T, T, xb.data_ptr(), T, st), 5, 10)
    out["blanker_k6"] = {"config": "K6 alone: CNoiseProc::ProcessBlanker (threshold 50, width 2 us) on %d receivers x 2^%d samples" % (C, T.bit_length() - 1),
                         "kernel": "csdr::noiseblank_kernel", "ms_per_launch": round(ms, 4),
                         "MSamples_per_s": round(C * T / ms / 1e3, 1),
                         "roofline": roofline_obj(C * T * 16.0 / ms / 1e6, ms, "csdr::noiseblank_kernel", C * T * 16.0, None),
                         "frac_of_hbm_peak": round(C * T * 16.0 / ms / 1e6 / HBM_PEAK_GBS, 4), "cpu_baseline": None}
    if check:                                             # fresh object, two calls (history across them), sample-exact
        import numpy as np
        from oracle import oracle as orc
        c = C // 3
        nb2 = ca.NoiseProcBatch(C, device=ctx.local)
        nb2.setup(True, 50.0, 2.0, C4_FS)
        q = orc.CNoiseProc(); q.SetupBlanker(True, 50.0, 2.0, C4_FS)
        xc = to_c128(x[c])
        same, blanked = True, 0
        for _ in range(2):
            nb2.process_ptr(x.data_ptr(), T, T, xb.data_ptr(), T, st)
            torch.cuda.synchronize()
            got = xb[c].cpu().numpy()
            want = np.asarray(q.ProcessBlanker(xc)).astype(np.complex64)
            same = same and bool(np.array_equal(got[:, 0] + 1j * got[:, 1], want))
            blanked += int((want == 0).sum())
        out["blanker_k6"]["parity_checked"] = {"receiver": c, "samples": int(2 * T), "samples_blanked": blanked, "sample_exact": same, "ok": same}
        del nb2
    del xb, nb
    if with_cpu:
        from oracle import oracle as orc
        xs = to_c128(x[1, :19968 * 16])
        d = orc.CDownConvert()
        d.SetDataRate(C4_FS, 15000.0); d.SetFrequency(-100.5e3)
        v, n = cpu_rate(lambda: [d.ProcessData(xs[i:i + 19968]) for i in range(0, len(xs), 19968)], len(xs), 2.0)
        out["downconv_k2"]["cpu_baseline"] = cpu_obj(v, n, "CDownConvert::ProcessData, 19968-sample calls, receiver 1's stream")
        q = orc.CNoiseProc()
        q.SetupBlanker(True, 50.0, 2.0, C4_FS)
        v, n = cpu_rate(lambda: [q.ProcessBlanker(xs[i:i + 256]) for i in range(0, len(xs), 256)], len(xs), 2.0)
        out["blanker_k6"]["cpu_baseline"] = cpu_obj(v, n, "CNoiseProc::ProcessBlanker, 256-sample calls (one datagram each)")
    return out


def blank_mask_call(torch, ca, nb, pk, npk, stream):
    """csdr__noiseproc_batch_mask (the blanker pass csdr_demod_batch_process_packets runs in front of the fused
    down-converters) on a datagram buffer, alone: returns a callable and the mask buffer it fills."""
    import ctypes as C_
    L = ca.lib()
    L.csdr__noiseproc_batch_mask.restype = C_.c_int
    L.csdr__noiseproc_batch_mask.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_longlong, C_.c_void_p, C_.c_int, C_.c_int, C_.c_int,
                                             C_.c_void_p, C_.c_longlong, C_.POINTER(C_.c_void_p), C_.POINTER(C_.c_void_p),
                                             C_.c_void_p]
    C, n = pk.shape[0], npk * 240
    words = (n + 31) // 32 + 64
    mask = torch.empty((C, words), device=pk.device, dtype=torch.int32)
    st, hi = C_.c_void_p(), C_.c_void_p()

    def run():
        rc = L.csdr__noiseproc_batch_mask(nb.h, None, 0, C_.c_void_p(pk.data_ptr()), npk, 1444, n, C_.c_void_p(mask.data_ptr()),
                                          words, C_.byref(st), C_.byref(hi), C_.c_void_p(stream))
        assert rc == 0, ca._capi.last_error()
    return run, mask


def datagrams_of(torch, x, npk):
    """the receivers' own signals in the radio's 24-bit wire format: value * 256, 3 LE bytes, 240 samples + 4 header
    bytes per datagram (interface/netiobase.cpp:479-527)"""
    C = x.shape[0]
    pk = torch.zeros((C, npk, 1444), device=x.device, dtype=torch.uint8)
    for c0 in range(0, C, 32):
        v = torch.round(x[c0:c0 + 32, :npk * 240].reshape(-1, npk, 480) * 256.0).clamp(-(1 << 23), (1 << 23) - 1).to(torch.int32)
        body = torch.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], dim=-1).to(torch.uint8).reshape(-1, npk, 1440)
        pk[c0:c0 + 32, :, 4:] = body
        del v, body
    return pk


def packets_chain(torch, ca, ctx, c4, check=True):
    """The C4 share fed with the radio's own 24-bit datagrams (interface/netiobase.cpp:479-527: 6 B per sample instead
    of 8) -- no unpack pass, the down-converter decodes them in its loads -- and the same with CNoiseProc's blanker in
    front (interface/sdrinterface.cpp:884), FUSED: the blanker kernel leaves one bit per sample, the down-converter takes
    the delayed sample itself and zeroes it under the mask (no blanked copy of the input is written).  Strict mode."""
    C, T, x = c4.C, c4.T, c4.x
    npk = (T // 240) // 8 * 8                            # 1920 = 64 * 30 samples: a multiple of the largest decimation
    Tp = npk * 240
    pk = datagrams_of(torch, x, npk)
    c4.set_mode(False)
    nb = ca.NoiseProcBatch(C, device=ctx.local)
    nb.setup(True, 50.0, 2.0, C4_FS)
    out = {"config": "C4 share from 24-bit datagrams: %d receivers x %d samples per call, strict mode" % (C, Tp)}
    for key, blk in (("packets_chain_ms", None), ("packets_blanker_chain_ms", nb)):
        def run():
            rc = ca.lib().csdr_demod_batch_process_packets(c4.b.h, pk.data_ptr(), npk, 1444, blk.h if blk is not None else None,
                                                           c4.aud.data_ptr(), c4.cap, c4.stream)
            assert rc == 0, ca._capi.last_error()
        out[key] = round(gpu_ms(torch, run, 8, 20), 4)
    out["raw_input_MSamples_per_s"] = round(C * Tp / out["packets_chain_ms"] / 1e3, 1)
    out["with_blanker_MSamples_per_s"] = round(C * Tp / out["packets_blanker_chain_ms"] / 1e3, 1)
    out["blanker"] = ("fused: noiseblank_kernel<mask, ring> (one bit per sample; the 5 ms window's magnitudes in an LDS ring) + "
                      "downconv_kernel<plan, BLK> (CSDR_BLANK_FUSED=0: two passes through a blanked fp32 copy)")
    # the mask kernel of that chain alone, roofline-shaped: datagram bytes in, one bit per sample out
    run, mask = blank_mask_call(torch, ca, nb, pk, npk, c4.stream)
    ms = gpu_ms(torch, run, 5, 20)
    alg = float(C) * (npk * 1444 + Tp / 8.0)
    out["mask_kernel"] = {"config": "the blanker pass of that chain alone: %d receivers x %d datagrams in, one bit per sample out" % (C, npk),
                          "ms_per_launch": round(ms, 4),
                          "roofline": roofline_obj(alg / ms / 1e6, ms, "csdr::noiseblank_kernel<true, true>", alg, None)}
    # the 16-bit wire format (256 samples + 4 header bytes per datagram): the same chain with the blanker
    npk16 = (T // 256) // 8 * 8
    pk16 = torch.zeros((C, npk16, 1028), device=x.device, dtype=torch.uint8)
    for c0 in range(0, C, 32):
        v = torch.round(x[c0:c0 + 32, :npk16 * 256].reshape(-1, npk16, 512)).clamp(-32768, 32767).to(torch.int32)
        pk16[c0:c0 + 32, :, 4:] = torch.stack([v & 255, (v >> 8) & 255], dim=-1).to(torch.uint8).reshape(-1, npk16, 1024)
        del v
    nb16 = ca.NoiseProcBatch(C, device=ctx.local)
    nb16.setup(True, 50.0, 2.0, C4_FS)
    def run16():
        rc = ca.lib().csdr_demod_batch_process_packets(c4.b.h, pk16.data_ptr(), npk16, 1028, nb16.h, c4.aud.data_ptr(), c4.cap, c4.stream)
        assert rc == 0, ca._capi.last_error()
    out["packets16_blanker_chain_ms"] = round(gpu_ms(torch, run16, 8, 20), 4)
    del nb16, pk16
    if check:
        out["parity_checked"] = {"datagrams": c4.parity_check(packets=(pk, npk)),
                                 "datagrams_with_blanker": c4.parity_check(packets=(pk, npk), blanker=True)}
        out["parity_checked"]["ok"] = bool(out["parity_checked"]["datagrams"]["ok"] and out["parity_checked"]["datagrams_with_blanker"]["ok"])
    del nb, pk, mask
    return out


def chain_one_receiver(torch, ca, ctx, with_cpu, name, check=True):
    """BASELINE configs C2 (2 MSPS -> CDownConvert -> 16384-pt CFastFIR -> AGC -> FM) and C5 (10 MSPS -> ... 2048-pt
    filter -> FM -> CFractResampler to 48 kHz): ONE receiver resident in HBM.  A single receiver cannot fill the
    chip: these are latency-bound rates, reported as such (no roofline fraction is claimed for them)."""
    dev = torch.device("cuda", ctx.local)
    if name == "c2":
        fs, fc, nfft, T = 2e6, 100e3, 16384, 1 << 23
        label = "C2: 1 receiver, 2 MSPS -> CDownConvert -> 16384-pt CFastFIR -> AGC -> FM, 2^23 raw samples per call"
    else:
        fs, fc, nfft, T = 10e6, 1.2e6, 2048, 1 << 24
        label = ("C5: 1 receiver, 10 MSPS -> CDownConvert (78 125 S/s) -> 2048-pt CFastFIR -> AGC -> FM -> "
                 "CFractResampler to 48 kHz, 2^24 raw samples per call")
    x = fm_stream(torch, dev, T, fs, fc)
    b = ca.DemodBatch(1, nfft, device=ctx.local)
    b.set_input_rate(fs); b.set_demod(0, ca.DEMOD_FM, fm_defaults(ca)); b.commit(); b.set_freq(0, -fc)
    out_rate = b.output_rate(0)
    dec = int(round(fs / out_rate))
    cap = T // dec + nfft + 4096
    aud = torch.zeros((1, cap), device=dev, dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    rs = ca.ResamplerBatch(1, device=ctx.local) if name == "c5" else None
    rate = out_rate / 48000.0
    pcm = torch.zeros((1, int(cap / rate) + 64), device=dev, dtype=torch.float32) if rs else None
    n_aud = (T // dec // 1024) * 1024                            # whole hops: what every call delivers in steady state

    def step():
        b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), cap, st)
        if rs:
            rs.resample_ptr(aud.data_ptr(), cap, n_aud, rate, pcm.data_ptr(), pcm.shape[1], None, st)
    ms = gpu_ms(torch, step, 4, 8)
    out = {"config": label, "ms_per_call": round(ms, 3), "raw_input_MSamples_per_s": round(T / ms / 1e3, 1),
           "x_real_time": round(T / ms / 1e3 / (fs / 1e6), 1), "output_rate": out_rate,
           "bound": "latency of one receiver's sequential stages (one workgroup walks its bursts)", "cpu_baseline": None}
    if check:
        # the buffer that was just timed, through a FRESH object (the timed one has seen it a dozen times), against the
        # oracle's CDemodulator on the same samples -- the call length (2^23 / 2^24) is not a multiple of m_InBufLimit:
        # the oracle holds back its last partial window, the common prefix is compared
        from oracle import oracle as orc
        fb = ca.DemodBatch(1, nfft, device=ctx.local)
        fb.set_input_rate(fs); fb.set_demod(0, ca.DEMOD_FM, fm_defaults(ca)); fb.commit(); fb.set_freq(0, -fc)
        aud.zero_()
        fb.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), cap, st)
        torch.cuda.synchronize()
        got = aud[0, :fb.out_count(0)].cpu().numpy()
        ro = orc.CDemodulator(nfft)
        ro.SetInputSampleRate(fs); ro.SetDemod(orc.DEMOD_FM, fm_defaults(orc)); ro.SetDemodFreq(-fc)
        want = ro.process_append(to_c128(x[0]))
        out["parity_checked"] = dict(chain_burst_check(got, want, "FM", hop=nfft // 2), receiver="the timed buffer, one call of 2^%d samples "
                                     "through a fresh object vs oracle CDemodulator::ProcessData" % (T.bit_length() - 1))
        if rs:
            rs2 = ca.ResamplerBatch(1, device=ctx.local)
            k = len(want) // 1024 * 1024
            rs2.resample_ptr(aud.data_ptr(), cap, k, rate, pcm.data_ptr(), pcm.shape[1], None, st)
            torch.cuda.synchronize()
            q = orc.CFractResampler(); q.Init(k + 64)           # one call, like the timed step: the same output times
            ref = q.Resample(want[:k], rate)
            pg = pcm[0, :len(ref)].cpu().numpy()
            out["parity_checked"]["resampled"] = chain_burst_check(pg, ref, "FM")
        del fb
    if with_cpu:
        from oracle import oracle as orc
        r = orc.CDemodulator(nfft)
        r.SetInputSampleRate(fs); r.SetDemod(orc.DEMOD_FM, fm_defaults(orc)); r.SetDemodFreq(-fc)
        lim = r.buf_limit()
        xs = to_c128(x[0, :lim * 40])
        q = orc.CFractResampler() if name == "c5" else None
        if q:
            q.Init(8192)
        def once():
            a = r.process_append(xs)
            if q:
                for j in range(0, len(a), 1024):
                    q.Resample(a[j:j + 1024], rate)
        v, n = cpu_rate(once, len(xs), 2.5)
        out["cpu_baseline"] = cpu_obj(v, n, "CDemodulator::ProcessData%s on the same stream, m_InBufLimit windows"
                                      % (" + CFractResampler::Resample" if q else ""))
    del b, rs, x, aud
    return out


def host_form(ca, with_cpu, check=True):
    """The path the reference's Qt host takes through the drop-in CDemodulator: complex DOUBLES in host memory, handed
    over in calls of 256 samples (one datagram each, interface/sdrinterface.cpp:903), audio back in host memory --
    2 MSPS, FM defaults, 2048-point filter.  PCIe, the fp64 <-> fp32 conversions and every launch are inside; one
    receiver is latency-bound by construction.  Never `value`.  The oracle's CDemodulator on the same samples, one host
    core, is timed beside it, and the audio of the timed samples is compared with it under the chain rule."""
    import ctypes as C
    import numpy as np
    from cutesdr_amd import _capi
    fs, fc, n, call = 2e6, 100e3, 1 << 22, 256
    t = np.arange(n) / fs
    rng = np.random.default_rng(5)
    x = 3276.7 * np.exp(1j * (2 * np.pi * fc * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t))) + 10.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x = np.ascontiguousarray(x, dtype=np.complex128)
    d = ca.CDemodulator(2048)
    d.SetInputSampleRate(fs); d.SetDemod(ca.DEMOD_FM, fm_defaults(ca)); d.SetDemodFreq(-fc)
    L = _capi.lib()
    L.csdr__demod_process_calls.restype = C.c_int
    L.csdr__demod_process_calls.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    out = np.zeros(n // 16 + 65536)
    run = lambda: _capi.check(L.csdr__demod_process_calls(d.h, n, call, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "host form")
    k_first = run()                                       # also the warm-up: staging buffers sized, clocks up
    first = out[:k_first].copy()
    reps, t0 = 3, time.perf_counter()
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / reps
    res = {"config": "drop-in CDemodulator, host doubles in / out: 2 MSPS FM, %d-sample calls (the reference's call pattern), 2^22 samples per measurement" % call,
           "raw_input_MSamples_per_s": round(n / dt / 1e6, 1), "x_real_time": round(n / dt / fs, 1),
           "staging": "pinned windows used in turn, read by the down-converter itself over PCIe (zero copy, round 5), audio written by the post-chain into a pinned buffer; one wait per pass that returns audio",
           "cpu_baseline": None}
    if with_cpu or check:
        from oracle import oracle as orc
        r = orc.CDemodulator(2048)
        r.SetInputSampleRate(fs); r.SetDemod(orc.DEMOD_FM, fm_defaults(orc)); r.SetDemodFreq(-fc)
        t0 = time.perf_counter()
        want = r.process_append(x)
        t_first = time.perf_counter() - t0
        if check:
            res["parity_checked"] = dict(chain_burst_check(first, want, "FM"), samples=int(n), counts_equal=bool(k_first == len(want)))
        if with_cpu:
            v, m = cpu_rate(lambda: r.process_append(x), n, 2.0)
            res["cpu_baseline"] = cpu_obj(v, m, "CDemodulator::ProcessData on the same stream")
    del d
    # the same with deferred output (csdr_demod_set_deferred, opt-in): a pass hands over the previous pass's audio, the
    # chain's pass runs while the host converts the next window; the audio must be the undeferred object's, word for word
    d = ca.CDemodulator(2048)
    d.SetInputSampleRate(fs); d.SetDemod(ca.DEMOD_FM, fm_defaults(ca)); d.SetDemodFreq(-fc)
    d.set_deferred(True)
    run = lambda: _capi.check(L.csdr__demod_process_calls(d.h, n, call, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "host form, deferred")
    k1 = run()
    late = np.concatenate([out[:k1].copy(), d.flush()])
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    dt2 = (time.perf_counter() - t0) / reps
    res["deferred_output"] = {"raw_input_MSamples_per_s": round(n / dt2 / 1e6, 1),
                              "same_words_one_window_later": bool(len(late) == len(first) and np.array_equal(late, first))}
    del d
    return res


# ---------------------------------------------------------------- timing helpers
def timed_steps(torch, ctx, step, steps, warmup, prewarm=True):
    """W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize on both sides; returns
    (elapsed seconds, MAX over ranks; mean HIP-event duration of a step on this rank in ms)."""
    if prewarm:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < PREWARM_S:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dist_barrier(ctx)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    dist_barrier(ctx)
    torch.cuda.synchronize()
    elapsed = dist_max(ctx, time.perf_counter() - t0)
    return elapsed, sum(a.elapsed_time(b) for a, b in ev) / steps


def _read_power_w(pci=None):
    """socket power in watts of the device at PCI address `pci` ("dddd:bb:dd.f"): its amdgpu hwmon node (no process
    started), else one rocm-smi call (a host shows every card's node: the first one found is not ours)"""
    import glob
    for name in ("power1_average", "power1_input"):
        for f in sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*/%s" % (pci or "*", name))):
            try:
                v = float(open(f).read().strip()) * 1e-6
                if v > 1.0:
                    return v, "hwmon"
            except (OSError, ValueError):
                pass
    try:
        import re
        out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        for card in json.loads(out).values():
            for k, v in card.items():
                if "ower" in k:
                    m = re.search(r"[0-9.]+", str(v))
                    if m:
                        return float(m.group(0)), "rocm-smi"
    except Exception:
        pass
    return None, None


def power_and_clock(torch, ca, ctx, step, seconds=1.5):
    """Socket power and the shader clock the chip holds WHILE the headline launch repeats (after the timed region, never
    inside it): the kernel sits at the socket's power cap, so a box that lands under the target says why in
    its own record.  Clock: one-wave probes (csdr__clock_probe: shader cycles per 100 MHz tick) on a side stream beside
    the launches; power: the hwmon node read every 20 ms (or rocm-smi).  The side stream is BORROWED from the library's
    stream pool and handed back (a torch stream of the bench's own took one of the process's hardware queues for good: the
    chain measured after it ran 1.98 instead of 1.61 ms; probes in the launch stream itself read the IDLE clock, 2.34 GHz --
    the chip raises its clock within microseconds of the load ending)."""
    import ctypes as C
    import threading
    L = ca.lib()
    L.csdr__clock_probe.restype = C.c_int
    L.csdr__clock_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    pr = torch.cuda.get_device_properties(ctx.local)
    pci = None
    if getattr(pr, "pci_bus_id", None) is not None:
        pci = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0) or 0, pr.pci_bus_id, getattr(pr, "pci_device_id", 0) or 0)
    nwg, rounds = 8, []
    samples, src, stop = [], [None], [False]

    def sampler():
        while not stop[0]:
            v, how = _read_power_w(pci)
            if v is not None:
                samples.append(v); src[0] = how
            time.sleep(0.02 if how == "hwmon" else 0.2)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        buf = torch.zeros((2 * nwg,), device="cuda", dtype=torch.int64)
        for i in range(60):
            step()
            if i == 30 and L.csdr__clock_probe(ctx.local, None, C.c_void_p(buf.data_ptr()), nwg, 2000) != 0:
                break
        torch.cuda.synchronize()
        rounds.append(buf.cpu().numpy().reshape(nwg, 2).astype(float))
    stop[0] = True
    th.join(timeout=2.0)
    import numpy as np
    clk = None
    if rounds:
        a = np.concatenate(rounds[1:] or rounds)
        ok = a[:, 1] > 0
        if ok.any():
            clk = float(np.median(a[ok, 0] / a[ok, 1])) * 0.1          # cycles per 10 ns tick -> GHz
    tail = samples[len(samples) // 3:]
    return {"power_w": round(sum(tail) / len(tail), 1) if tail else None, "power_source": src[0], "power_samples": len(tail),
            "sclk_ghz": None if clk is None else round(clk, 3), "pci": pci}


# ---------------------------------------------------------------- workloads
class C3Workload:
    """256 channels x 2^19 samples through the 16384-pt overlap-save filter, all resident in HBM."""

    def __init__(self, torch, ca, ctx, channels, blocks_per_wg=0):
        self.torch, self.ca, self.C, self.T = torch, ca, channels, T_PER_CH
        lo, _ = shard_channels(ctx, channels * ctx.world)      # this rank's channels: lo .. lo+C-1
        dev = torch.device("cuda", ctx.local)
        g = torch.Generator(device=dev)
        g.manual_seed(0xC0DE0000 + lo)
        # synthetic IQ resident in HBM: noise at -20 dBFS of a 16-bit full scale
        self.x = torch.randn((self.C, self.T, 2), generator=g, device=dev, dtype=torch.float32) * 3276.7
        self.y = torch.empty_like(self.x)
        self.fir = ca.FastFirBatch(self.C, FFT_N, device=ctx.local)
        self.fir.setup(-5000, 5000, 0, FS)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.bpw = blocks_per_wg

    def step(self):
        self.fir.process_ptr(self.x.data_ptr(), self.T, self.T, self.y.data_ptr(), self.T, self.stream, self.bpw)

    def parity_check(self, nch=4):
        """The buffer that was just timed against the fp64 oracle: the stream is periodic (the same input every
        step), so the last step's output equals the oracle's second pass over the same samples."""
        import numpy as np
        from oracle import oracle as orc
        self.torch.cuda.synchronize()
        chans = sorted(set(int(c) for c in np.linspace(0, self.C - 1, nch)))
        worst = 0.0
        for c in chans:
            xc = self.x[c].cpu().numpy().astype(np.float64)
            xc = xc[:, 0] + 1j * xc[:, 1]
            ff = orc.CFastFIR(FFT_N)
            ff.SetupParameters(-5000, 5000, 0, FS)
            ff.ProcessData(xc)
            ref = ff.ProcessData(xc)
            yc = self.y[c].cpu().numpy().astype(np.float64)
            worst = max(worst, float(np.abs((yc[:, 0] + 1j * yc[:, 1]) - ref).max() / np.abs(xc).max()))
        return {"channels": chans, "samples_each": self.T, "max_err_over_max_abs_x": worst, "tolerance": 2e-5,
                "ok": bool(worst <= 2e-5)}

    def other_sizes(self, ctx, steps=60, check=True):
        """The same launch at CFastFIR's other sizes (dsp/fastfir.cpp:55-56 as a parameter: 2048 is the reference's own,
        what every receiver of the chain runs), 16 B per sample as the headline, and the single-pass display spectrum
        sizes on the same buffer (8 B per bin read)."""
        torch = self.torch
        out = {"fastfir": {}, "spectrum": {}}
        n = self.C * self.T
        for size in (2048, 4096, 8192):
            fir = self.ca.FastFirBatch(self.C, size, device=ctx.local)
            fir.setup(-5000, 5000, 0, FS)
            ms = gpu_ms(torch, lambda: fir.process_ptr(self.x.data_ptr(), self.T, self.T, self.y.data_ptr(), self.T, self.stream, 0), 40, steps)
            out["fastfir"][str(size)] = {"ms": round(ms, 4), "GBps_at_16B_per_sample": round(16.0 * n / ms / 1e6, 1),
                                         "frac": round(16.0 * n / ms / 1e6 / HBM_PEAK_GBS, 4)}
            if check:                                  # the stream is periodic: the last launch = the oracle's second pass
                import numpy as np
                from oracle import oracle as orc
                torch.cuda.synchronize()
                c = self.C // 3
                xc = self.x[c].cpu().numpy().astype(np.float64)
                xc = xc[:, 0] + 1j * xc[:, 1]
                ff = orc.CFastFIR(size); ff.SetupParameters(-5000, 5000, 0, FS)
                ff.ProcessData(xc)
                ref = ff.ProcessData(xc)
                yc = self.y[c].cpu().numpy().astype(np.float64)
                err = float(np.abs((yc[:, 0] + 1j * yc[:, 1]) - ref).max() / np.abs(xc).max())
                out["fastfir"][str(size)]["parity_checked"] = {"channel": c, "samples": self.T, "max_err_over_max_abs_x": err,
                                                               "tolerance": 2e-5, "ok": bool(err <= 2e-5)}
            del fir
        for size in (2048, 4096, 8192, 16384):
            fb = self.ca.FftBatch(self.C, device=ctx.local)
            fb.set_params(size, False, 0.0, C4_FS); fb.set_ave(1)
            # (60 untimed launches first: the parity spot check in between leaves the GPU idle long enough to drop its clocks)
            ms = gpu_ms(torch, lambda: fb.put_display_ptr(self.x.data_ptr(), self.T, self.T // size, self.stream), 60, 40)
            out["spectrum"][str(size)] = {"ms": round(ms, 4), "GBps_at_8B_per_bin": round(8.0 * n / ms / 1e6, 1),
                                          "frac": round(8.0 * n / ms / 1e6 / HBM_PEAK_GBS, 4)}
            if check:                                  # no averaging: the display holds the call's LAST frame
                import numpy as np
                from oracle import oracle as orc
                torch.cuda.synchronize()
                c = self.C // 3
                last = self.x[c, self.T - size:].cpu().numpy().astype(np.float64)
                r = orc.CFft(); r.SetFFTParams(size, False, 0.0, C4_FS); r.SetFFTAve(1)
                r.PutInDisplayFFT(last[:, 0] + 1j * last[:, 1])
                want = np.asarray(r.ave_buf())
                got = fb.ave_buf(c).astype(np.float64)
                near = want > want.max() - 6.0         # the rule of tests/test_fft_resampler_gpu.py: 0.01 dB within 60 dB of the peak
                err = float(np.abs(got - want)[near].max())
                out["spectrum"][str(size)]["parity_checked"] = {"channel": c, "bins_within_60dB_of_peak": int(near.sum()),
                                                                "max_err_bels": err, "tolerance_bels": 0.001, "ok": bool(err <= 0.001)}
            del fb
        out["note"] = ("%d channels x 2^%d samples per launch; fastfir 16384 is the headline; kernels: fastfir_os2_kernel<11> / <12> / "
                       "<13> at 2048 / 4096 / 8192 points (round 5: the headline kernel's pipelined build at every size); spectrum 2048 / 4096 / 8192 at sixteen points per thread (2048 and "
                       "8192: round 4), 16384 generic" % (self.C, self.T.bit_length() - 1))
        return out

    def distinct_filters(self, ctx, steps=100):
        """Same launch with one H per channel (pass-bands staggered by 10 Hz): +16 B/sample of filter reads."""
        torch = self.torch
        fir = self.ca.FastFirBatch(self.C, FFT_N, device=ctx.local)
        for c in range(self.C):
            fir.setup(-5000 + 10 * (c % 32), 5000 + 10 * (c % 32), 0, FS, channel=c)
        def step():
            fir.process_ptr(self.x.data_ptr(), self.T, self.T, self.y.data_ptr(), self.T, self.stream, self.bpw)
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        n = self.C * self.T
        return {"kernel_ms": round(ms, 4), "MSamples_per_s": round(n / ms / 1e3, 1),
                "achieved_GBps_at_16B_per_sample": round(16.0 * n / ms / 1e6, 1),
                "GBps_counting_the_H_reads": round(32.0 * n / ms / 1e6, 1),
                "frac_at_16B_per_sample": round(16.0 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
                "note": "every channel its own filter: N*8 B of H per N/2-sample hop = +16 B/sample of filter reads (SURVEY "
                        "8d); a channel's 128 KB of H is re-read by one workgroup for 64 consecutive hops, so after the "
                        "first hop those reads are served by L2 / Infinity Cache, not HBM"}


class C4Workload:
    """Per-GPU share of BASELINE config C4: mixed AM / FM / USB receivers, each tuned to its own carrier in its
    own 2 MSPS stream, through csdr_demod_batch (CDemodulator::ProcessData, dsp/demodulator.cpp:163-215)."""

    def __init__(self, torch, ca, ctx, channels):
        self.torch, self.ca, self.ctx, self.C, self.T = torch, ca, ctx, channels, C4_T
        dev = torch.device("cuda", ctx.local)
        lo, _ = shard_channels(ctx, channels * ctx.world)
        g = torch.Generator(device=dev)
        g.manual_seed(0xC0DE0000 + lo)
        A = 3276.7                                                # -20 dBFS carrier, AWGN -70 dBFS per component
        C, T = self.C, self.T
        x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * (32767.0 * 10 ** (-70 / 20))
        t = torch.arange(T, device=dev, dtype=torch.float64) / C4_FS
        for c in range(C):
            gc = lo + c                                           # global channel id: kind and carrier follow it
            fc = 100e3 + 500.0 * (gc % 1024)
            if gc % 3 == 0:                                       # AM 50 % / 1 kHz
                ph = 2 * torch.pi * fc * t
                amp = A * (1.0 + 0.5 * torch.sin(2 * torch.pi * 1000.0 * t))
                x[c, :, 0] += (amp * torch.cos(ph)).float(); x[c, :, 1] += (amp * torch.sin(ph)).float()
            elif gc % 3 == 1:                                     # FM +-3 kHz / 1 kHz
                ph = 2 * torch.pi * fc * t + 3.0 * torch.sin(2 * torch.pi * 1000.0 * t)
                x[c, :, 0] += (A * torch.cos(ph)).float(); x[c, :, 1] += (A * torch.sin(ph)).float()
            else:                                                 # two-tone SSB
                for off in (1200.0, 2340.0):
                    ph = 2 * torch.pi * (fc + off) * t
                    x[c, :, 0] += (0.5 * A * torch.cos(ph)).float(); x[c, :, 1] += (0.5 * A * torch.sin(ph)).float()
        del t
        self.x = x
        base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                    FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                    AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
        modes = [(ca.DEMOD_AM, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)),
                 (ca.DEMOD_FM, dict()),
                 (ca.DEMOD_USB, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200,
                                     LowCutmin=0, Symetric=0))]
        def make_batch(pipelined):
            b = ca.DemodBatch(C, 2048, device=ctx.local)
            b.set_input_rate(C4_FS)
            for c in range(C):
                m, kw = modes[(lo + c) % 3]
                b.set_demod(c, m, ca.DemodInfo(**dict(base, **kw)))
            b.commit()
            for c in range(C):
                b.set_freq(c, -(100e3 + 500.0 * ((lo + c) % 1024)))
            if pipelined:
                b.set_pipelined(True)    # streaming host: the post-chain of step k overlaps the down-converter of step k+1
            return b
        self.make_batch = make_batch
        # the batch object is made on first use, in the mode asked for: a strict-mode measurement must not inherit the
        # hardware queues a pipelined object's nine streams left behind (1.85 against 2.1-2.3 ms: what a host that never
        # asks for pipelining gets is the former)
        self.mode, self.b = None, None
        self.default_pipelined = not os.environ.get("CSDR_BENCH_STRICT_CHAIN")
        self.cap = T // 16 + 4096                                  # audio row capacity (highest output rate: /32)
        self.aud = torch.zeros((C, self.cap), device=dev, dtype=torch.float32)
        self.sm = torch.zeros((C,), device=dev, dtype=torch.float32)
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self):
        if self.b is None:
            self.set_mode(self.default_pipelined)
        self.b.process_ptr(self.x.data_ptr(), self.T, self.T, self.aud.data_ptr(), self.cap, self.stream)

    def gather(self, reps=5):
        """Per-channel S-meter and audio of one step gathered to rank 0 over RCCL (SURVEY sections 5, 8e): the
        only exchange the multi-channel configuration has.  Timed on its own, never part of `value`."""
        torch, ctx = self.torch, self.ctx
        n_aud = self.T // 32                                       # what every channel is guaranteed to have written
        payload = torch.empty((self.C, 1 + n_aud), device=self.aud.device, dtype=torch.float32)
        def pack():
            self.b.smeter_all_ptr(self.sm.data_ptr(), None, self.stream)
            payload[:, 0] = self.sm
            payload[:, 1:] = self.aud[:, :n_aud]
        pack()
        torch.cuda.synchronize()
        if not getattr(ctx, "dist_on", ctx.world > 1):
            return {"ms": None, "bytes_to_rank0": 0, "note": "single rank: nothing to gather",
                    "smeter_db_first4": [round(float(v), 2) for v in self.sm[:4].cpu()]}
        on_gpu = ctx.backend == "nccl"                            # (gloo rehearsal: through host memory)
        wire = payload if on_gpu else payload.cpu()
        # 64 receivers per message: 17 MB from every rank, 4 messages per step -- bounds rank 0's staging to
        # world x 17 MB per message in flight; RCCL would cut one 67 MB message into pieces of its own anyway
        rows = 64
        parts = [wire[r0:r0 + rows] for r0 in range(0, self.C, rows)]
        dsts = [[torch.empty_like(p) for _ in range(ctx.world)] if ctx.rank == 0 else None for p in parts]
        def gather_all():
            for p, d in zip(parts, dsts):
                ctx.dist.gather(p, d, dst=0)
        gather_all()                                               # warm (communicator setup)
        torch.cuda.synchronize()
        dist_barrier(ctx)
        t0 = time.perf_counter()
        for _ in range(reps):
            pack()
            if not on_gpu:
                wire.copy_(payload)
            gather_all()
        torch.cuda.synchronize()
        dist_barrier(ctx)
        ms = dist_max(ctx, time.perf_counter() - t0) / reps * 1e3
        nbytes = payload.numel() * 4 * (ctx.world - 1)
        return {"ms": round(ms, 3), "bytes_to_rank0": nbytes, "GBps_into_rank0": round(nbytes / ms / 1e6, 1),
                "backend": ctx.dist.get_backend(), "messages_per_step": len(parts),
                "collective": "torch.distributed.gather (RCCL send/recv over xGMI) in messages of 64 receivers, "
                              "S-meter + %d audio samples per receiver" % n_aud}

    def control_plane(self, ctx, steps=12):
        """What the control plane costs (VERDICT r5 task 6): between two calls of the PIPELINED 256-receiver batch, every
        receiver is retuned (csdr_demod_batch_set_freq) and given new filter edges (a same-mode SetDemod: new frequency
        response, AGC constants checked, squelch / AM low-pass re-set) -- 256 + 256 calls, nothing waited for.
        retune_us = the host time of those 512 calls; step_increase_ms = how much longer a step takes with them in front
        of every step (the patch kernels read 256 x 2 x 16 KB of responses from pinned memory) than without.
        parity_ok: the same calls setting what is ALREADY set (FM / USB receivers; an AM SetDemod clears its low-pass as the
        reference's does) leave every audio word of the next steps as it was."""
        torch, ca = self.torch, self.ca
        self.set_mode(True)
        b, C = self.b, self.C
        lo, _ = shard_channels(ctx, C * ctx.world)
        base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                    FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                    AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
        modes = [(ca.DEMOD_AM, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)),
                 (ca.DEMOD_FM, dict()),
                 (ca.DEMOD_USB, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0))]

        split = {"freq": 0.0, "demod": 0.0, "n": 0}

        def touch(step, same):
            t0 = time.perf_counter()
            for c in range(C):
                m, kw = modes[(lo + c) % 3]
                kw = dict(base, **kw)
                if not same:
                    kw["HiCut"] = kw["HiCut"] - 100 * (1 + step % 2)          # new edges every step
                    info = ca.DemodInfo(**kw)
                    ta = time.perf_counter()
                    b.set_freq(c, -(100e3 + 500.0 * ((lo + c) % 1024)) - 10.0 * (step % 2))
                    tb = time.perf_counter()
                    b.set_demod(c, m, info)
                    tc = time.perf_counter()
                    split["freq"] += tb - ta; split["demod"] += tc - tb; split["n"] += 1
                elif m != ca.DEMOD_AM:
                    b.set_freq(c, -(100e3 + 500.0 * ((lo + c) % 1024)))
                    b.set_demod(c, m, ca.DemodInfo(**kw))
            return time.perf_counter() - t0

        def run(retune, same=False):
            for _ in range(4):
                self.step()
            b.flush(self.stream); torch.cuda.synchronize()
            host, t0 = 0.0, time.perf_counter()
            for k in range(steps):
                if retune:
                    host += touch(k, same)
                self.step()
            b.flush(self.stream); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3, host / steps * 1e6
        # parity first (the object is still at its original settings): three steps untouched, three with no-op setters
        def audio_of_three(touching):
            self.set_mode(True, fresh=True)                                 # a fresh object: identical start on both sides
            nonlocal b
            b = self.b
            rows = []
            for k in range(3):
                if touching and k > 0:
                    touch(k, True)
                self.step()
                b.flush(self.stream); torch.cuda.synchronize()
                rows.append(self.aud[1:C:3, :self.T // 32].clone())       # the FM receivers
                rows.append(self.aud[2:C:3, :self.T // 32].clone())       # the USB receivers
            return rows
        skip = os.environ.get("CSDR_CP_SKIP", "")               # (diagnostic: tools/experiments/r6_repro_mode3.py)
        plain, touched = ([], []) if "parity" in skip else (audio_of_three(False), audio_of_three(True))
        parity_ok = all(bool(torch.equal(p, t)) for p, t in zip(plain, touched))
        ms_plain, _ = (0.0, 0.0) if "plain" in skip else run(False)
        ms_retune, host_us = (0.0, 0.0) if "retune" in skip else run(True)
        return {"config": "pipelined batch, %d receivers: set_freq + same-mode set_demod (new filter edges) for EVERY receiver in front of every step" % C,
                "retune_us": round(host_us, 1), "calls_per_step": 2 * C,
                "set_freq_us_per_call": round(split["freq"] / max(1, split["n"]) * 1e6, 2),
                "set_demod_us_per_call": round(split["demod"] / max(1, split["n"]) * 1e6, 2), "ms_per_step_plain": round(ms_plain, 4),
                "ms_per_step_with_retunes": round(ms_retune, 4), "step_increase_ms": round(ms_retune - ms_plain, 4),
                "parity_ok": parity_ok, "parity_what": "no-op set_freq / set_demod on the FM and USB receivers: every audio word of three steps equal to an untouched batch's"}

    def set_mode(self, pipelined, fresh=False):
        """the batch object of the wanted mode: ONE per mode for the workload's life, made on first use (a strict-mode
        object never creates the pipelined mode's streams: that is the path a host that never asks for pipelining runs)
        and kept while the other mode is timed; fresh=True replaces it by a new one (identical start for a comparison).
        (Until round 6 every switch dropped the object and made a new one; what such a new object ran on depended on the
        order in which the library's stream pool got its streams back -- 1.6 or 2.0-2.2 ms per step for the object's whole
        life: tools/experiments/r6_repro_mode*.py, HISTORY.  The pool hands out the process's oldest streams now; the
        workload keeps its objects anyway.)"""
        want = "pipelined" if pipelined else "strict"
        if want == self.mode and not fresh:
            return
        if self.b is not None:
            self.b.flush(self.stream)
            self.torch.cuda.synchronize()
        if not hasattr(self, "kept"):
            self.kept = {}
        if self.mode in ("strict", "pipelined") and self.b is not None:
            self.kept[self.mode] = self.b
        self.b = None
        if fresh and want in self.kept:
            del self.kept[want]
            import gc
            gc.collect()
        self.b = self.kept.pop(want) if want in self.kept else self.make_batch(pipelined)
        self.mode = want

    def parity_check(self, receivers=(0, 1, 2), packets=None, blanker=False):
        """The buffer that was just timed through a FRESH batch object of all the receivers, one call of 2^21 samples
        (not a multiple of m_InBufLimit = 19968), receivers 0-2 (AM, FM, USB) against the oracle's CDemodulator on the
        same samples -- the chain rule of the parity tests, burst by burst, on the common prefix (the oracle holds back
        its last partial window).  packets = (datagram tensor, datagrams per call): the same through
        csdr_demod_batch_process_packets, with CNoiseProc's blanker in front on both sides when blanker is set (the FM
        bounds then start one burst later: impulses into an empty delay line)."""
        from oracle import oracle as orc
        import numpy as np
        torch = self.torch
        b = self.make_batch(False)
        aud = torch.zeros_like(self.aud)
        if packets is None:
            b.process_ptr(self.x.data_ptr(), self.T, self.T, aud.data_ptr(), self.cap, self.stream)
        else:
            pk, npk = packets
            nb = self.ca.NoiseProcBatch(self.C, device=self.ctx.local) if blanker else None
            if nb is not None:
                nb.setup(True, 50.0, 2.0, C4_FS)
            rc = self.ca.lib().csdr_demod_batch_process_packets(b.h, pk.data_ptr(), npk, 1444, nb.h if nb is not None else None,
                                                              aud.data_ptr(), self.cap, self.stream)
            assert rc == 0, self.ca._capi.last_error()
        torch.cuda.synchronize()
        base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                    FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                    AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
        modes = [("AM", orc.DEMOD_AM, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)), ("FM", orc.DEMOD_FM, dict()),
                 ("USB", orc.DEMOD_USB, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0))]
        lo, _ = shard_channels(self.ctx, self.C * self.ctx.world)
        res, ok = [], True
        for c in receivers:
            name, m, kw = modes[(lo + c) % 3]
            r = orc.CDemodulator(2048)
            r.SetInputSampleRate(C4_FS); r.SetDemod(m, orc.DemodInfo(**dict(base, **kw)))
            r.SetDemodFreq(-(100e3 + 500.0 * ((lo + c) % 1024)))
            if packets is None:
                xin = to_c128(self.x[c])
            else:
                xin = np.asarray(orc.unpack_packets(packets[0][c].cpu().numpy(), 1444), dtype=np.complex128)
                if blanker:
                    q = orc.CNoiseProc(); q.SetupBlanker(True, 50.0, 2.0, C4_FS)
                    xin = np.asarray(q.ProcessBlanker(xin))
            want = r.process_append(xin)
            got = aud[c, :b.out_count(c)].cpu().numpy()
            chk = dict(chain_burst_check(got, want, name, fm_late=1 if blanker else 0), receiver=c, mode=name)
            ok = ok and chk["ok"]
            res.append(chk)
        del b, aud
        return {"receivers": res, "samples_per_receiver": self.T, "ok": ok,
                "what": "the timed buffer, one call through a fresh strict-mode object vs oracle CDemodulator::ProcessData"}

    def cpu_baseline(self, budget_s=4.0):
        """the oracle's CDemodulator on ONE host core over the first three receivers' own streams (AM, FM, USB by
        turns, like the shard), m_InBufLimit windows: raw input samples per second and core"""
        from oracle import oracle as orc
        base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                    FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                    AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
        modes = [(orc.DEMOD_AM, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)), (orc.DEMOD_FM, dict()),
                 (orc.DEMOD_USB, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0))]
        lo, _ = shard_channels(self.ctx, self.C * self.ctx.world)
        chains, streams = [], []
        for c in range(3):
            m, kw = modes[(lo + c) % 3]
            r = orc.CDemodulator(2048)
            r.SetInputSampleRate(C4_FS); r.SetDemod(m, orc.DemodInfo(**dict(base, **kw)))
            r.SetDemodFreq(-(100e3 + 500.0 * ((lo + c) % 1024)))
            chains.append(r)
            streams.append(to_c128(self.x[c, :19968 * 16]))
        def once():
            for r, xs in zip(chains, streams):
                r.process_append(xs)
        v, n = cpu_rate(once, 3 * len(streams[0]), budget_s)
        return cpu_obj(v, n, "CDemodulator::ProcessData (dsp/demodulator.cpp:163-215) on receivers 0-2 of the shard")

    def summary(self, ctx, steps, warmup, with_cpu=False, check=True):
        """both modes of csdr_demod_batch: `pipelined` (successive calls overlapped, a call's results are complete one
        call later: what a streaming host uses) is the headline of this object, `strict` (complete in stream order
        when the call returns) is reported beside it"""
        n = self.C * self.T
        def one(pipelined):
            self.set_mode(pipelined)
            elapsed, ms = timed_steps(self.torch, ctx, self.step, steps, warmup, prewarm=False)
            return {"mode": self.mode, "ms_per_step": round(elapsed / steps * 1e3, 4), "event_ms_per_step": round(ms, 4),
                    "raw_input_MSamples_per_s_all_gpus": round(n * steps * ctx.world / elapsed / 1e6, 1),
                    "algorithmic_GBps_per_gpu": round((8.0 + 4.0 / 32.0) * n / ms / 1e6, 1),
                    "frac_of_hbm_peak": round((8.0 + 4.0 / 32.0) * n / ms / 1e6 / HBM_PEAK_GBS, 4)}
        strict = one(False)
        out = {"channels_per_gpu": self.C, "raw_samples_per_channel": self.T, "steps": steps,
               "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}    # (the pipelined mode needs more than HIP's default 4)
        out.update(one(True))
        out["strict"] = strict
        alg = (8.0 + 4.0 / 32.0) * n
        out["roofline"] = roofline_obj(alg / strict["event_ms_per_step"] / 1e6, strict["event_ms_per_step"],
                                       "whole chain, strict mode: every launch of one csdr_demod_batch_process call", alg, None)
        if check and ctx.rank == 0:
            out["parity_checked"] = self.parity_check()
        out["gather"] = self.gather()
        out["cpu_baseline"] = self.cpu_baseline() if with_cpu else None
        return out


class StubWorkload:
    """CPU stand-in for the tests of the multi-rank path (no GPU, gloo): sleeps instead of launching."""

    def __init__(self, channels):
        self.C, self.T = channels, 1 << 12

    def step(self):
        time.sleep(0.002)


def run_stub(args):
    ctx = dist_init(backend="gloo")
    w = StubWorkload(args.channels)
    lo, hi = shard_channels(ctx, args.channels * ctx.world)
    for _ in range(args.warmup):
        w.step()
    dist_barrier(ctx)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.step()
    dist_barrier(ctx)
    elapsed = dist_max(ctx, time.perf_counter() - t0)
    ranges = [[lo, hi]]
    if ctx.world > 1:                                  # every rank's channel range, for the tests of the sharding rule
        ranges = [None] * ctx.world
        ctx.dist.all_gather_object(ranges, [lo, hi])
    if ctx.rank == 0:
        line = result_line(ctx, w.C, w.T, args.steps, args.warmup, elapsed, elapsed / args.steps * 1e3,
                           extra={"stub": True, "rank0_channels": [lo, hi], "rank_channels": ranges})
        print(json.dumps(line), flush=True)
    dist_finish(ctx)


def run_rank(args):
    # the chain forks its decimator-plan groups onto several streams; HIP's default of 4 hardware queues would
    # serialise some of them (read when the runtime initialises, so before torch touches the GPU)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libcutesdr_mi has no CPU fallback)")
    ctx = dist_init()
    torch.cuda.set_device(ctx.local)
    import cutesdr_amd as ca

    extra = {}
    if args.workload == "c3":
        w = C3Workload(torch, ca, ctx, args.channels, args.blocks_per_wg)
        elapsed, kern_ms = timed_steps(torch, ctx, w.step, args.steps, args.warmup)
        if ctx.rank == 0 and not args.no_check:
            extra["parity_checked"] = w.parity_check()
        if ctx.rank == 0 and ctx.world == 1 and not args.no_secondary:
            extra["power_clock"] = power_and_clock(torch, ca, ctx, w.step)
        if ctx.rank == 0 and not args.no_secondary:
            extra["distinct_filters"] = w.distinct_filters(ctx)
            if ctx.world == 1:
                extra["fastfir_sizes"] = w.other_sizes(ctx, check=not args.no_check)
        chans, samples = w.C, w.T
        del w
        torch.cuda.empty_cache()
        if not args.no_secondary:
            with_cpu = ctx.world == 1 and not args.no_cpu       # CPU legs: rank 0 at N = 1 only
            c4 = C4Workload(torch, ca, ctx, CHANNELS)
            s = c4.summary(ctx, 30, 15, with_cpu, check=not args.no_check)
            if ctx.rank == 0:
                extra["chain_c4"] = s
            if ctx.world == 1:                                   # single-GPU configurations: not part of a scaling run
                # (the datagram chain on the strict object chain_c4 has just timed; the control plane, which makes and drops
                # objects, last)
                extra["packets_chain"] = packets_chain(torch, ca, ctx, c4, check=not args.no_check)
                extra["spectrum_c1"] = spectrum_c1(torch, ca, ctx, c4.x, with_cpu, check=not args.no_check)
                extra.update(input_rate_kernels(torch, ca, ctx, c4.x, with_cpu, check=not args.no_check))
                extra["control_plane"] = c4.control_plane(ctx)
            del c4
            torch.cuda.empty_cache()
            if ctx.world == 1:
                extra["host_form"] = host_form(ca, with_cpu, check=not args.no_check)
                extra["chain_c2"] = chain_one_receiver(torch, ca, ctx, with_cpu, "c2", check=not args.no_check)
                extra["chain_c5"] = chain_one_receiver(torch, ca, ctx, with_cpu, "c5", check=not args.no_check)
    else:
        w = C4Workload(torch, ca, ctx, args.channels)
        elapsed, kern_ms = timed_steps(torch, ctx, w.step, args.steps, args.warmup)
        g = w.gather()
        if ctx.rank == 0:
            extra["gather"] = g
            extra["chain_mode"] = w.mode
        chans, samples = w.C, w.T
        del w
    census = rank_census(ctx)
    if ctx.rank == 0:
        extra["ranks"] = census

    if ctx.rank == 0:
        traffic = None
        if args.workload == "c3":
            live = live_traffic() if (ctx.world == 1 and not args.no_secondary and not os.environ.get("CSDR_BENCH_NO_PMC")) else None
            if live and live.get("k1"):
                traffic = live["k1"]["bytes"]
                extra["traffic_measured"] = dict({k: v for k, v in live["k1"].items() if k != "bytes"}, source=live["source"])
                # the secondary objects' rooflines get their counter traffic from the same two passes
                for key, name in (("k2", "downconv_k2"), ("k3", "spectrum_c1"), ("k6", "blanker_k6"), ("chain", "chain_c4")):
                    if live.get(key) and name in extra and "roofline" in extra[name]:
                        r = extra[name]["roofline"]
                        r["traffic"] = live[key]["bytes"]
                        r["traffic_over_algorithmic"] = round(live[key]["bytes"] / r["algorithmic_bytes_per_launch"], 3)
                        r["traffic_counters_KiB"] = {k: v for k, v in live[key].items() if k not in ("bytes", "GB_by_kernel")}
                        if live[key].get("GB_by_kernel"):
                            r["traffic_GB_by_kernel"] = live[key]["GB_by_kernel"]
                if live.get("k6m") and "mask_kernel" in extra.get("packets_chain", {}):
                    r = extra["packets_chain"]["mask_kernel"]["roofline"]
                    r["traffic"] = live["k6m"]["bytes"]
                    r["traffic_over_algorithmic"] = round(live["k6m"]["bytes"] / r["algorithmic_bytes_per_launch"], 3)
                    r["traffic_counters_KiB"] = {k: v for k, v in live["k6m"].items() if k != "bytes"}
            else:
                traffic = profiled_traffic()
                extra["traffic_measured"] = {"source": "profiles/traffic_latest.json (same kernel sources: hash checked)" if traffic else None}
        cpu = cpu_baseline() if (ctx.world == 1 and not args.no_cpu) else None
        print(json.dumps(result_line(ctx, chans, samples, args.steps, args.warmup, elapsed, kern_ms, traffic, cpu,
                                     args.workload, extra)), flush=True)
    dist_finish(ctx)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", choices=("c3", "c4"), default="c3")
    ap.add_argument("--channels", type=int, default=CHANNELS, help="channels per GPU")
    ap.add_argument("--blocks-per-wg", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-check", action="store_true", help="skip the post-timing parity spot check")
    ap.add_argument("--no-secondary", action="store_true", help="skip the distinct-filter and chain measurements")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)     # tests: CPU stand-in workload, gloo
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the counter passes of live_traffic()
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves, before this process imports torch or touches the GPU
        return spawn_ranks(args.gpus, argv)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus and int(os.environ.get("RANK", "0")) == 0:
        print("bench.py: --gpus %d but the launcher started %s ranks; using the launcher's" %
              (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
    if args.pmc_child:
        run_pmc_child()
    elif args.stub:
        run_stub(args)
    else:
        run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
