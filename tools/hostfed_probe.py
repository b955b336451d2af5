#!/usr/bin/env python
"""dev helper (GPU box): how well does a pinned H2D copy overlap the forward?  one 36.5 MB block vs 21 tensors, copy stream vs none."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

m = GraphNet(2, 64)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in shipped_state().items()})
eng = m.engine()
dev = eng.device
batch = synth.make_batch("cifar_base_kw", 256, seed=1234)
args = batch.forward_args()
d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
d[4], d[6] = args[4].to(dev), args[6].to(dev)
for _ in range(10):
    eng.forward(*d)
torch.cuda.synchronize()
n = 36473856 // 4
pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
dst = torch.empty(n, dtype=torch.float32, device=dev)
cs = torch.cuda.Stream(device=dev)
K = 20


def timeit(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


print("forward alone            %.3f ms" % timeit(lambda: eng.forward(*d)))
def copy_only():
    with torch.cuda.stream(cs):
        dst.copy_(pin, non_blocking=True)
print("one 36.5 MB copy alone   %.3f ms" % timeit(copy_only))
def both():
    with torch.cuda.stream(cs):
        dst.copy_(pin, non_blocking=True)
    eng.forward(*d)
print("copy (side stream) + fwd %.3f ms" % timeit(both))
chunks = torch.chunk(pin, 21)
dchunks = torch.chunk(dst, 21)
def both21():
    with torch.cuda.stream(cs):
        for a, b in zip(dchunks, chunks):
            a.copy_(b, non_blocking=True)
    eng.forward(*d)
print("21 copies (side) + fwd   %.3f ms" % timeit(both21))
tiny = [torch.empty(256, dtype=torch.float32, pin_memory=True) for _ in range(8)]
dtiny = [torch.empty(256, dtype=torch.float32, device=dev) for _ in range(8)]
def both21_tiny():
    with torch.cuda.stream(cs):
        for a, b in zip(dchunks, chunks):
            a.copy_(b, non_blocking=True)
        for a, b in zip(dtiny, tiny):
            a.copy_(b, non_blocking=True)
    eng.forward(*d)
print("21 copies + 8 tiny + fwd %.3f ms" % timeit(both21_tiny))
mid = [torch.empty(25600, dtype=torch.float32, pin_memory=True) for _ in range(4)]
dmid = [torch.empty(25600, dtype=torch.float32, device=dev) for _ in range(4)]
def both21_mid():
    with torch.cuda.stream(cs):
        for a, b in zip(dchunks, chunks):
            a.copy_(b, non_blocking=True)
        for a, b in zip(dmid, mid):
            a.copy_(b, non_blocking=True)
    eng.forward(*d)
print("21 copies + 4 x 100 KB + fwd %.3f ms" % timeit(both21_mid))
ev = torch.cuda.Event()
def both21_events():
    with torch.cuda.stream(cs):
        for a, b in zip(dchunks, chunks):
            a.copy_(b, non_blocking=True)
        ev.record(cs)
    torch.cuda.current_stream().wait_event(ev)
    eng.forward(*d)
print("21 copies, forward waits for them (no overlap by construction) %.3f ms" % timeit(both21_events))
