"""The N>1 path of bench.py on the CPU: two gloo ranks shard the channels (no data-path
collective), synchronise with a barrier and reduce the elapsed time with MAX, exactly the
torch.distributed calls bench.py makes on RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert abs(out["tmax"] - 0.15) < 1e-9                      # MAX over ranks, not rank 0's own time
    assert out["range0"] == [0, 1024]                          # contiguous channel ranges per rank
    line = out["line"]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["value"] == round(2 * 1024 * (1 << 12) * 4 / 0.15 / 1e6, 2)     # whole-job aggregate
    assert line["cpu_baseline"] is None
    for k in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in line
