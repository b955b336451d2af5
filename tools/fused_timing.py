#!/usr/bin/env python
"""dev helper (GPU box): per-phase cycle shares of k_gather_update, one fused half-pass at a time.
   Needs a library built with -DFUSED_TIMING (GNNB_LIB=...): runs a forward with every half-pass but one on the two-kernel path
   is not possible, so it reports the sums over all fused launches of one forward, split by resetting between forwards of
   networks / batch sizes given on the command line:  python tools/fused_timing.py cifar_base_kw 256"""
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
names = ["staging", "decode+bounds wait", "slot table", "walk", "compaction", "chain", "last partial tile", "-", "-"]
tot = sum(out[i] for i in range(9))
waves = out[15]
print(f"{net} B={B}: {waves // n} waves per forward (all fused launches), {tot / max(waves, 1):.0f} cycles per wave")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {100.0 * out[i] / tot:5.1f} %   {out[i] / max(waves, 1):9.0f} cycles per wave")
