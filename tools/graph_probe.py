#!/usr/bin/env python
"""dev helper (GPU box): does replaying the forward's launches from a captured HIP graph shorten the gaps between its dependent kernels?
   python tools/graph_probe.py [net] [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

net = sys.argv[1] if len(sys.argv) > 1 else "cifar_base_kw"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
m = GraphNet(2, 64)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in shipped_state().items()})
eng = m.engine()
batch = synth.make_batch(net, B, seed=1234)
args = batch.forward_args()
d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
d[4], d[6] = args[4].to(dev), args[6].to(dev)


def timed(label, fn, n=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{label:40s} {1e3 * (time.perf_counter() - t0) / n:.4f} ms per forward", flush=True)


timed("stream launches", lambda: eng.forward(*d))
ref = eng.forward(*d)
torch.cuda.synchronize()
want = ref.scores.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        eng.forward(*d)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res = eng.forward(*d)
torch.cuda.synchronize()
res.scores.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay reproduces the scores bit for bit:", bool(torch.equal(res.scores, want)), " status", res.status.tolist())
timed("graph replay", g.replay)
timed("stream launches", lambda: eng.forward(*d))
timed("graph replay", g.replay)
