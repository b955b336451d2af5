#!/usr/bin/env python
"""dev helper (GPU box): throughput of TWO independent batches in flight (two handles, two streams, full-size batches) against one."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

net = sys.argv[1] if len(sys.argv) > 1 else "cifar_base_kw"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
engs, argsets = [], []
for k in range(2):
    m = GraphNet(2, 64)
    m.load_state_dict({n: torch.as_tensor(np.asarray(v)) for n, v in shipped_state().items()})
    engs.append((m, m.engine()))
    batch = synth.make_batch(net, B, seed=1234 + k)
    a = batch.forward_args()
    d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in a]
    d[4], d[6] = a[4].to(dev), a[6].to(dev)
    argsets.append(d)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
K = 200


def one_stream():
    for i in range(K):
        engs[0][1].forward(*argsets[0])


def two_streams():
    for i in range(K):
        with torch.cuda.stream(streams[i & 1]):
            engs[i & 1][1].forward(*argsets[i & 1])


for name, f in (("one batch at a time", one_stream), ("two batches in flight", two_streams), ("one batch at a time", one_stream), ("two batches in flight", two_streams)):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    print(f"{net} B={B}: {name:24s} {1e3 * (time.perf_counter() - t0) / K:.4f} ms per batch", flush=True)
r0 = engs[0][1].forward(*argsets[0]); torch.cuda.synchronize(); r0.check()
