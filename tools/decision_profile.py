#!/usr/bin/env python
"""dev helper (GPU box): cProfile of GraphChoice.decision (B=1, host tensors) -- where the host time of one call goes."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_score import GraphChoice
ckpt = os.path.join(ROOT, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
batch = synth.make_batch("cifar_base_kw", 1, seed=3)
init_mask = [m[0] for m in batch.bab_masks]
g = GraphChoice(init_mask, ckpt); g.verbose = False
args = (batch.lower_bounds_all, batch.upper_bounds_all, batch.dual_vars, batch.primal_inputs, list(batch.primals), batch.layers, init_mask)
for _ in range(50): g.decision(*args)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): g.decision(*args)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("NROWS", "14")))
