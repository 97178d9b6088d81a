#!/usr/bin/env python3
"""GPU box, round 6: does a strict object made after pipelined ones run at the speed of a fresh one?  (bench.py's order:
chain_c4 pipelined + strict, control_plane -- several pipelined objects with retunes --, then the datagram chain on a new
strict object)"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)

def remake(w, pipelined):
    """drop every batch object the workload holds and make a new one (what C4Workload.set_mode did on every switch until
    round 6)"""
    import gc
    if w.b is not None:
        w.b.flush(w.stream); torch.cuda.synchronize()
    w.b = None; w.kept = {}; w.mode = None
    gc.collect()
    w.set_mode(pipelined)
out = {}
def t(label):
    out[label] = round(bench.gpu_ms(torch, w.step, 8, 20), 3)
def pk(label):
    p = bench.packets_chain(torch, ca, ctx, w, check=False); out[label] = (p["packets_chain_ms"], p["packets_blanker_chain_ms"])
remake(w, False); t("strict_fresh"); pk("packets_fresh")
remake(w, True); t("pipelined")
remake(w, False); t("strict_after_pipelined"); pk("packets_after_pipelined")
cp = w.control_plane(ctx); out["control_plane_step_increase_ms"] = cp.get("step_increase_ms")
remake(w, False); t("strict_after_control_plane"); t("strict_after_control_plane_again"); pk("packets_after_control_plane")
t("strict_after_control_plane_third")
remake(w, False); t("another_strict_object")
remake(w, True); t("another_pipelined_object")
remake(w, False); t("and_a_strict_object_behind_it")
print(json.dumps(out))
