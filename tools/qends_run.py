#!/usr/bin/env python
"""dev helper (GPU box): GNNB_LIB=tools/ablate/qends.so python3 tools/qends_run.py cifar_base_kw 256
One forward; per variant of k_gather_update_q (its LAST launch of the forward): when the workgroups' gather waves and chain waves finished,
relative to the first workgroup's start -- how much of the kernel is the wait for its slowest workgroup."""
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
out = (C.c_ulonglong * (4 * 3 * 512))()
assert lib.gnnb_dev_qends(out, 1) == 0
m.forward_device(*args)
torch.cuda.synchronize()
assert lib.gnnb_dev_qends(out, 1) == 0
a = np.array(out[:], dtype=np.uint64).reshape(4, 3, 512)
names = {0: "<16, dense src>", 1: "<16, sparse src>", 2: "<16, embedding>", 3: "<32, transposed>"}
for v in range(4):
    ge, ce = a[v, 1].astype(np.int64), a[v, 2].astype(np.int64)
    ok = ge > 0
    if not ok.any():
        continue
    g, c = ge[ok] * 0.01, ce[ok] * 0.01
    pct = lambda x: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
    print(f"{net} B={B} {names[v]}: {ok.sum()} workgroups, time from a workgroup's own start (us; a variant launched more than once: the longer of its launches)")
    print("   gather waves done ", pct(g))
    print("   chain waves done  ", pct(c))
