#!/usr/bin/env python
"""dev helper: index and length of every slow eng.forward() call in a bench-like sequence (pre-warm un-synced, sync, rest)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
sd = torch.load(os.path.join(ROOT, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt"), map_location="cpu", weights_only=True)
m = GraphNet(2, 64); m.load_state_dict(sd); eng = m.engine()
net, B, pre = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
batch = synth.make_batch(net, B, seed=1234)
dev = torch.device("cuda")
dl = lambda ts: [t.to(dev).float().contiguous() for t in ts]
args = (dl(batch.lower_bounds_all), dl(batch.upper_bounds_all), dl(batch.dual_vars), dl(batch.primals), batch.primal_inputs.to(dev), batch.layers, batch.masks.to(dev))
t_start = time.perf_counter()
slow = []
def fwd(i):
    a = time.perf_counter(); eng.forward(*args); d = time.perf_counter() - a
    if d > 3e-3: slow.append((i, round(1e3 * d, 1), round(1e3 * (a - t_start), 1)))
i = 0
for _ in range(pre): fwd(i); i += 1
torch.cuda.synchronize()
for rep in range(6):
    t0 = time.perf_counter()
    for _ in range(23): fwd(i); i += 1
    torch.cuda.synchronize()
    print(f"rep {rep} (forwards {i-23}..{i-1}): {1e3*(time.perf_counter()-t0)/23:.3f} ms/forward")
print("slow calls (index, ms, at ms):", slow)
