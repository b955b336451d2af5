#!/usr/bin/env python
"""dev helper (GPU box): does the overlap of pinned H2D copies with the forward depend on WHICH side stream carries them?
Creates side streams one after the other and times {18 copies of 2 MB on stream k} + {forward on the current stream} for each."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(20):
    eng.forward(*d)
torch.cuda.synchronize()
P = 1 << 19
pin = torch.empty(18 * P, dtype=torch.float32, pin_memory=True)
dst = torch.empty(18 * P, dtype=torch.float32, device=dev)
K = 40


def timeit(f):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


print("forward alone %.3f ms" % timeit(lambda: eng.forward(*d)), flush=True)
streams = []
for k in range(12):
    s = torch.cuda.Stream(device=dev)
    streams.append(s)
    ev = torch.cuda.Event()

    def both():
        with torch.cuda.stream(s):
            for j in range(18):
                dst[j * P:(j + 1) * P].copy_(pin[j * P:(j + 1) * P], non_blocking=True)
        eng.forward(*d)
    print("side stream #%2d (handle %#x): copies + forward %.3f ms" % (k, s.cuda_stream, timeit(both)), flush=True)

# ---- does it matter where the destination / the pinned source come from?
import gc
s = streams[0]


def run(tag, dst_, pin_):
    def both():
        with torch.cuda.stream(s):
            for j in range(18):
                dst_[j * P:(j + 1) * P].copy_(pin_[j * P:(j + 1) * P], non_blocking=True)
        eng.forward(*d)
    print("%-60s %.3f ms" % (tag, timeit(both)), flush=True)


run("same buffers again", dst, pin)
del dst
gc.collect()
torch.cuda.empty_cache()
dst2 = torch.empty(18 * P, dtype=torch.float32, device=dev)
run("device buffer allocated after empty_cache()", dst2, pin)
pin2 = torch.empty(18 * P, dtype=torch.float32).pin_memory()
run("pinned source from .pin_memory() of a pageable tensor", dst2, pin2)
del pin
gc.collect()
pin3 = torch.empty(18 * P, dtype=torch.float32, pin_memory=True)
run("pinned source allocated after the first one was freed", dst2, pin3)
x = [torch.randn(1 << 20).pin_memory() for _ in range(20)]
del x
gc.collect()
pin4 = torch.randn(18 * P).pin_memory()
run("pinned source from the host allocator's cache", dst2, pin4)
