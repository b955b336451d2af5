#!/usr/bin/env python
"""dev helper: host time of eng.forward() over many repetitions -- shows the one-off pauses of Python's cyclic GC (a full
collection takes 35-100 ms with torch loaded) that bench.py keeps out of its timed region."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
sd = torch.load(os.path.join(ROOT, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt"), map_location="cpu", weights_only=True)
m = GraphNet(2, 64); m.load_state_dict(sd); eng = m.engine()
net = sys.argv[2] if len(sys.argv) > 2 else "cifar_base_kw"
batch = synth.make_batch(net, int(sys.argv[3]) if len(sys.argv) > 3 else 256, seed=1234)
dev = torch.device("cuda")
dl = lambda ts: [t.to(dev).float().contiguous() for t in ts]
args = (dl(batch.lower_bounds_all), dl(batch.upper_bounds_all), dl(batch.dual_vars), dl(batch.primals), batch.primal_inputs.to(dev), batch.layers, batch.masks.to(dev))
for _ in range(5): eng.forward(*args)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(12):
    t0 = time.perf_counter()
    worst = 0.0
    for _ in range(n):
        a = time.perf_counter(); eng.forward(*args); worst = max(worst, time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: enqueue {1e3*(t1-t0)/n:.3f} ms/forward (worst call {1e3*worst:.2f} ms), total {1e3*(t2-t0)/n:.3f} ms/forward", flush=True)
