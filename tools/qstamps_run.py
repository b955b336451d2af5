#!/usr/bin/env python
"""dev helper (GPU box): GNNB_LIB=tools/ablate/qstamps.so python3 tools/qstamps_run.py cifar_base_kw 256
Prints, per variant of k_gather_update_q, what an average gather wave and an average chain wave spent their time on (us)."""
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
out = (C.c_ulonglong * 64)()
assert lib.gnnb_dev_qstamps(out, 1) == 0
n = 10
for _ in range(n):
    m.forward_device(*args)
torch.cuda.synchronize()
assert lib.gnnb_dev_qstamps(out, 1) == 0
names = {0: "<16, dense src>", 1: "<16, sparse src>", 2: "<16, embedding>", 3: "<32, transposed>"}
T = 0.01   # us per tick
for v in range(4):
    q = out[16 * v:16 * v + 16]
    if not q[6]:
        continue
    gw, cw = q[6], max(q[12], 1)
    print(f"{net} B={B} {names[v]}: {gw // n} gather-wave runs and {cw // n} chain-wave runs per forward")
    print(f"  gather wave (us): prologue {T*q[0]/gw:6.2f}  fetch/decode/bounds {T*q[1]/gw:6.2f}  table build {T*q[2]/gw:6.2f}  k-loop {T*q[3]/gw:6.2f}"
          f"  ring push {T*q[4]/gw:6.2f}  | alive {T*q[5]/gw:6.2f}  tiles {q[7]/gw:5.1f}")
    print(f"  chain wave  (us): staging {T*q[8]/cw:6.2f}  waiting for rows {T*q[9]/cw:6.2f}  chain {T*q[10]/cw:6.2f}  | alive {T*q[11]/cw:6.2f}  tiles {q[13]/cw:5.1f}")
