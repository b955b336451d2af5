#!/usr/bin/env python
"""dev helper (GPU box): steady-state latency of GraphChoice.decision through the reference's call surface (host tensors,
python-list primals, one subproblem), and of children_decisions (B=2)."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gnn_branching_amd import synth, bab_caller
from gnn_branching_amd.graphnet.graph_score import GraphChoice
ckpt = os.path.join(ROOT, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
for net in ("cifar_base_kw", "cifar_deep_kw"):
    batch = synth.make_batch(net, 2, seed=3)
    one = batch.slice(0, 1)
    init_mask = [m[0] for m in one.bab_masks]
    g = GraphChoice(init_mask, ckpt); g.verbose = False
    prim_lists = [p.tolist() for p in one.primals]
    args = (one.lower_bounds_all, one.upper_bounds_all, one.dual_vars, one.primal_inputs, prim_lists, one.layers, init_mask)
    for _ in range(20): g.decision(*args)
    gc.collect(); gc.disable()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); g.decision(*args); ts.append(time.perf_counter() - t0)
    gc.enable()
    ts.sort()
    print(f"{net}: GraphChoice.decision (B=1, host tensors + python lists): median {1e3*ts[100]:.3f} ms, p10 {1e3*ts[20]:.3f}, p90 {1e3*ts[180]:.3f}")
    # tensors instead of python lists for the primals
    args_t = (one.lower_bounds_all, one.upper_bounds_all, one.dual_vars, one.primal_inputs, list(one.primals), one.layers, init_mask)
    for _ in range(20): g.decision(*args_t)
    gc.collect(); gc.disable()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); g.decision(*args_t); ts.append(time.perf_counter() - t0)
    gc.enable()
    ts.sort()
    print(f"{net}: same with tensor primals: median {1e3*ts[100]:.3f} ms")
