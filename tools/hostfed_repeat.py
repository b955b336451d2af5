#!/usr/bin/env python
"""dev helper (GPU box): HostFedPipeline step time over repeated constructions, with and without gc / empty_cache in between."""
import gc, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_branching_amd import synth
from gnn_branching_amd.engine import HostFedPipeline
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

m = GraphNet(2, 64)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in shipped_state().items()})
eng = m.engine()
batch = synth.make_batch("cifar_base_kw", 256, seed=1234)
args = list(batch.forward_args())


def leg(tag, pinned=True):
    a = [[t.float().contiguous() for t in g] if isinstance(g, list) else g for g in args]
    a[4], a[6] = args[4].float().contiguous(), args[6].float().contiguous()
    if pinned:
        a = [[t.pin_memory() for t in g] if isinstance(g, list) else g for g in a]
        a[4], a[6] = a[4].pin_memory(), a[6].pin_memory()
    pipe = HostFedPipeline(eng)
    for _ in range(4):
        r = pipe.submit(*a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        r = pipe.submit(*a)
    torch.cuda.synchronize()
    print("%-44s %.3f ms per step" % (tag, 1e3 * (time.perf_counter() - t0) / 100), flush=True)


leg("first")
leg("second")
leg("pageable", pinned=False)
leg("third")
gc.collect()
leg("after gc.collect()")
torch.cuda.empty_cache()
leg("after empty_cache()")
gc.collect(); torch.cuda.empty_cache()
leg("after both")
leg("again")
if hasattr(torch._C, "_host_emptyCache"):
    gc.collect(); torch._C._host_emptyCache()
    leg("after emptying the pinned-host cache")
