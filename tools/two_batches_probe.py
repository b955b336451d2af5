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

# ---- the same through engine.BatchPipeline (what bench.py's side leg runs)
from gnn_branching_amd.engine import BatchPipeline
pipe = BatchPipeline(engs[0][0].state_dict(), depth=2)


def piped(sets):
    def f():
        keep = []
        for i in range(K):
            keep = (keep + [pipe.submit(*sets[i % len(sets)])])[-2:]
        pipe.synchronize()
    return f


for name, f in (("BatchPipeline, two different batches", piped(argsets)), ("BatchPipeline, the same batch twice", piped(argsets[:1])),
                ("two batches in flight (plain)", two_streams)):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    print(f"{net} B={B}: {name:40s} {1e3 * (time.perf_counter() - t0) / K:.4f} ms per batch", flush=True)

# ---- bisect: the pipeline's handles in the plain loop; the plain loop with an event per forward
def plain_with(engines, events):
    def f():
        for i in range(K):
            with torch.cuda.stream(streams[i & 1]):
                r = engines[i & 1].forward(*argsets[i & 1])
                if events:
                    e = torch.cuda.Event()
                    e.record(streams[i & 1])
    return f


for name, f in (("plain loop, the pipeline's handles", plain_with(pipe.engines, False)), ("plain loop, own handles, event per forward", plain_with([e[1] for e in engs], True)),
                ("plain loop, pipeline's handles and streams", None)):
    if f is None:
        def f():
            for i in range(K):
                with torch.cuda.stream(pipe.streams[i & 1]):
                    pipe.engines[i & 1].forward(*argsets[i & 1])
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    print(f"{net} B={B}: {name:46s} {1e3 * (time.perf_counter() - t0) / K:.4f} ms per batch", flush=True)

# ---- which PAIRS of streams overlap?
more = streams + list(pipe.streams) + [torch.cuda.Stream() for _ in range(4)] + [torch.cuda.Stream(priority=-1)]
own = [e[1] for e in engs]
for a in range(len(more)):
    for b in range(a + 1, len(more)):
        if not (b == a + 1 or a == 0):
            continue
        pair = (more[a], more[b])

        def f():
            for i in range(K):
                with torch.cuda.stream(pair[i & 1]):
                    own[i & 1].forward(*argsets[i & 1])
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        print(f"streams #{a} + #{b}{' (high priority)' if b == len(more) - 1 else ''}: {1e3 * (time.perf_counter() - t0) / K:.4f} ms per batch", flush=True)

# ---- the DEFAULT stream against side streams (what the RCCL all-gather of bench.py --gpus N meets: forwards on the default stream)
for b in range(len(more)):
    def f():
        for i in range(K):
            if i & 1:
                with torch.cuda.stream(more[b]):
                    own[1].forward(*argsets[1])
            else:
                own[0].forward(*argsets[0])
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    print(f"default stream + #{b}: {1e3 * (time.perf_counter() - t0) / K:.4f} ms per batch", flush=True)
