#!/usr/bin/env python3
"""dev helper (GPU box): average duration of EVERY launch of one forward, in launch order (HIP events through gnnb_profile_trace), and
the step time without instrumentation.   python3 tools/launch_trace.py [--net cifar_base_kw] [--batch 256] [--steps 50]
Environment knobs (GNNB_*, GNNB_LIB) apply as usual: run it twice for a same-box A/B of two settings."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--net", default="cifar_base_kw")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--tag", default="")
a = ap.parse_args()
from gnn_branching_amd import synth                                   # noqa: E402
from gnn_branching_amd.graphnet.graph_conv import GraphNet            # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sd = torch.load(os.path.join(ROOT, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt"), map_location="cpu", weights_only=True)
m = GraphNet(2, 64)
m.load_state_dict(sd)
eng = m.eval().engine()
dev = torch.device("cuda", 0)
b = synth.make_batch(a.net, a.batch, seed=1234)
args = b.forward_args()
d = [[t.to(dev).float().contiguous() for t in g] if isinstance(g, list) else g for g in args]
d[4], d[6] = args[4].to(dev), args[6].to(dev)
for _ in range(60):
    r = eng.forward(*d)
torch.cuda.synchronize()
r.check()
t0 = time.perf_counter()
for _ in range(a.steps):
    r = eng.forward(*d)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / a.steps
eng.profile_enable(True)
eng.profile_read(reset=True)
eng.profile_trace(65536)
for _ in range(a.steps):
    eng.forward(*d)
torch.cuda.synchronize()
eng.profile_read(reset=True)
tr = eng.profile_trace(65536)
eng.profile_enable(False)
n = len(tr) // a.steps
assert n * a.steps == len(tr), (len(tr), a.steps)
out = []
for i in range(n):
    names = {tr[i + j * n][0] for j in range(a.steps)}
    assert len(names) == 1
    out.append((tr[i][0], 1e3 * sum(tr[i + j * n][1] for j in range(a.steps)) / a.steps))
print(f"{a.tag or os.environ.get('GNNB_LIB', 'in-tree')} {a.net} B={a.batch}: {ms:.4f} ms/step, {n} launches, sum {sum(u for _, u in out):.1f} us: " +
      " ".join(f"{k.replace('k_', '')}={u:.1f}" for k, u in out))
