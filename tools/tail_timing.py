#!/usr/bin/env python
"""dev helper (GPU box): phase shares of k_scored_tail (library built with -DFUSED_TIMING=5, GNNB_LIB=...)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from gnn_branching_amd import _lib, synth  # noqa: E402
from gnn_branching_amd.graphnet.graph_conv import GraphNet  # noqa: E402
from tests.common import shipped_state  # noqa: E402

net, B = sys.argv[1], int(sys.argv[2])
m = GraphNet(2, 64)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in shipped_state().items()})
batch = synth.make_batch(net, B, seed=1234)
dev = torch.device("cuda")
args = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in batch.forward_args()]
args[4] = batch.primal_inputs.to(dev)
args[6] = batch.masks.to(dev)
lib = _lib.load()
for _ in range(3):
    m.forward_device(*args)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib.gnnb_debug_read(out, 1)
n = 5
for _ in range(n):
    m.forward_device(*args)
torch.cuda.synchronize()
lib.gnnb_debug_read(out, 1)
names = ["G start", "G gather", "G barrier wait", "G other layers' tiles", "G finish", "C staging / idle", "C barrier wait", "C last chain", "C other tiles", "C finish"]
pairs = out[15] / 2
print(f"{net} B={B}: {pairs / n:.0f} workgroups per launch (cycles per wave, 100 MHz s_memtime-equivalent shader clock)")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {out[i] / max(pairs, 1):9.0f} cycles")
