#!/usr/bin/env python
"""dev helper: whole-call rate when the caller hands over HOST tensors (the reference's call surface does): H2D copies of the
14 input tensors + forward, vs the device-resident rate bench.py reports."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
sd = torch.load(os.path.join(ROOT, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt"), map_location="cpu", weights_only=True)
m = GraphNet(2, 64); m.load_state_dict(sd); eng = m.engine()
B = 256
batch = synth.make_batch("cifar_base_kw", B, seed=1234)
host = batch.forward_args()
nbytes = sum(t.numel() * 4 for grp in host[:4] for t in grp) + host[4].numel() * 4 + host[6].numel() * 4
pinned = tuple([t.pin_memory() for t in g] if isinstance(g, list) else (g.pin_memory() if torch.is_tensor(g) else g) for g in host)
n_amb = int(batch.masks.sum())
for name, args in (("pageable", host), ("pinned", pinned)):
    for _ in range(3): eng.forward(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): eng.forward(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {1e3 * dt:.3f} ms per call incl. H2D of {nbytes / 1e6:.1f} MB -> {n_amb / dt / 1e6:.1f} M scores/s")
# the C-ABI host entry point (gnnb_forward_host: ONE pinned staging transfer each way, synchronous)
for _ in range(3): eng.forward_host(*host)
t0 = time.perf_counter()
for _ in range(10): eng.forward_host(*host)
dt = (time.perf_counter() - t0) / 10
print(f"gnnb_forward_host: {1e3 * dt:.3f} ms per call incl. staging + H2D of {nbytes / 1e6:.1f} MB and decisions back -> {n_amb / dt / 1e6:.1f} M scores/s")
