"""The N>1 path of bench.py on the CPU.  (1) bench.py --gpus 2 with no launcher starts two rank processes
itself (fresh children) and rank 0 prints ONE JSON line with n_gpus = 2; the ranks run a CPU stand-in
workload over gloo, everything else -- sharding, barrier, MAX-reduce, result line -- is the code the GPU run
uses.  (2) the same under torch.distributed.run, the way the driver launches it.  (3) the aggregation
arithmetic on known numbers."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_gpus_flag_spawns_the_ranks_itself():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "4",
                        "--warmup", "1", "--channels", "1024"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1                                     # rank 0 only, one line
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["rank0_channels"] == [0, 1024]                 # contiguous channel ranges, 1024 per rank
    # whole-job aggregate: both ranks' channels over the slowest rank's time
    assert abs(line["value"] - 2 * 1024 * 4096 * 4 / (line["ms_per_step"] * 4e-3) / 1e6) / line["value"] < 1e-3
    assert line["cpu_baseline"] is None and "roofline" in line


def test_single_rank_default_does_not_spawn():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1


def test_under_torch_distributed_run():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_eight_ranks_the_shape_of_the_scaling_run():
    """What the driver's 8-GPU run looks like to bench.py, rehearsed with eight gloo ranks on the CPU:
    started the driver's way (torch.distributed.run, 8 processes), eight CONTIGUOUS channel ranges of 256, ONE JSON line
    from rank 0, whole-job value = 8 x the per-rank work over the slowest rank's time."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert line["rank_channels"] == [[256 * k, 256 * (k + 1)] for k in range(8)]
    assert abs(line["value"] - 8 * 256 * 4096 * 3 / (line["ms_per_step"] * 3e-3) / 1e6) / line["value"] < 1e-3
    assert "x8" in line["config"]["parallelism"]


def test_two_rank_gloo_aggregation(tmp_path):
    script = tmp_path / "run.py"
    script.write_text(
        "import sys, json, time\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "ctx = bench.dist_init(backend='gloo')\n"
        "assert ctx.world == 2\n"
        "lo, hi = bench.shard_channels(ctx, 2048)\n"
        "bench.dist_barrier(ctx)\n"
        "elapsed = 0.10 + 0.05 * ctx.rank\n"
        "tmax = bench.dist_max(ctx, elapsed)\n"
        "line = bench.result_line(ctx, channels=hi - lo, samples=1 << 12, steps=4, warmup=1, elapsed=tmax, kern_ms=1.0)\n"
        "if ctx.rank == 0: print(json.dumps({'tmax': tmax, 'range0': [lo, hi], 'line': line}))\n"
        "bench.dist_finish(ctx)\n" % ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_lines(r.stdout)[-1]
    assert abs(out["tmax"] - 0.15) < 1e-9                      # MAX over ranks, not rank 0's own time
    assert out["range0"] == [0, 1024]                          # contiguous channel ranges per rank
    line = out["line"]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["value"] == round(2 * 1024 * (1 << 12) * 4 / 0.15 / 1e6, 2)     # whole-job aggregate
    assert line["cpu_baseline"] is None
    for k in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in line


import pytest


@pytest.mark.gpu
def test_two_ranks_rehearsed_on_one_gpu():
    """The GPU code of the multi-rank path -- both workloads, the S-meter/audio gather, the single JSON line --
    with two rank processes sharing cuda:0 and gloo in place of RCCL (CSDR_BENCH_ONE_GPU: a one-GPU box cannot
    host two RCCL ranks).  Started by bench.py's own --gpus spawn path."""
    env = dict(_clean_env(), CSDR_BENCH_ONE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-cpu"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["parity_checked"]["ok"]
    g = line["chain_c4"]["gather"]
    assert g["bytes_to_rank0"] == 256 * (1 + (1 << 21) // 32) * 4 and g["ms"] > 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4", "--steps", "3",
                        "--warmup", "2", "--no-cpu"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_lines(r.stdout)[0]
    assert line["n_gpus"] == 2 and "C4" in line["config"]["workload"] and line["gather"]["ms"] > 0


@pytest.mark.gpu
def test_one_rank_through_rccl():
    """VERDICT r5 task 3: RCCL itself, on the one GPU there is.  CSDR_BENCH_FORCE_DIST=1 makes bench.py build a ONE-rank
    "nccl" process group (a fresh child process, started before anything touches the GPU) and switches the world == 1
    short-cuts off: the barrier, the MAX all-reduce of the elapsed time on a DEVICE tensor, the all_gather_object census
    and the chain workload's torch.distributed.gather of S-meters and audio all go through librccl -- the same lines the
    8-GPU run executes.  N > 1 over xGMI stays unmeasured (no multi-GPU box is available to the builder)."""
    env = dict(_clean_env(), CSDR_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4", "--steps", "3", "--warmup", "2",
                        "--no-cpu"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_lines(r.stdout)[0]
    assert line["n_gpus"] == 1 and "C4" in line["config"]["workload"] and line["value"] > 0
    assert line["ranks"]["backend"] == "nccl" and line["ranks"]["world_size"] == 1
    assert line["ranks"]["ranks"][0]["device"] == 0
    g = line["gather"]
    assert g["backend"] == "nccl" and g["ms"] > 0 and g["messages_per_step"] == 4 and g["bytes_to_rank0"] == 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu", "--no-secondary"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_lines(r.stdout)[0]
    assert line["ranks"]["backend"] == "nccl" and line["parity_checked"]["ok"] and line["value"] > 0
