#!/usr/bin/env python
"""dev helper (GPU box): soak of engine.BatchPipeline -- N submits per configuration, alternating two batches, every result compared bit for
bit with the one-at-a-time forward and every status word checked."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.engine import BatchPipeline
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda")
for net, B in (("cifar_base_kw", 256), ("cifar_deep_kw", 128), ("cifar_wide_kw", 64), ("cifar_base_kw", 1), ("cifar_deep_kw", 2), ("cifar_base_kw", 17)):
    m = GraphNet(2, 64)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in shipped_state().items()})
    eng = m.engine()
    sets, want = [], []
    for seed in (21, 22):
        batch = synth.make_batch(net, B, seed=seed)
        a = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in batch.forward_args()]
        a[4] = batch.primal_inputs.to(dev); a[6] = batch.masks.to(dev)
        sets.append(a)
        with torch.no_grad():
            r = eng.forward(*a).check()
        want.append((r.scores.clone(), r.decisions.clone()))
    pipe = BatchPipeline(m.state_dict(), depth=2)
    t0 = time.time()
    pending = []
    with torch.no_grad():
        for i in range(N):
            pending.append((i % 3 % 2, pipe.submit(*sets[i % 3 % 2])))       # (0, 1, 0, 0, 1, 0, ...: both slots see both batches)
            if len(pending) >= 8:
                k, r = pending.pop(0)
                r.check()
                assert torch.equal(r.scores, want[k][0]) and torch.equal(r.decisions, want[k][1]), (net, B, i)
    for k, r in pending:
        r.check()
        assert torch.equal(r.scores, want[k][0]) and torch.equal(r.decisions, want[k][1]), (net, B)
    print(f"{net} B={B}: {N} submits through two slots identical to the plain forward, status clean ({time.time() - t0:.1f}s)", flush=True)
