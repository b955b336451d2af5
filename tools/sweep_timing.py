#!/usr/bin/env python3
"""dev helper (GPU box): where the workgroups of k_sweep are in time, phase by phase (library built with -DSWEEP_TIMING: tools/mklib.sh WORK sweept
-DSWEEP_TIMING; run with GNNB_LIB=tools/ablate/sweept.so GNNB_SWEEP=1).   python3 tools/sweep_timing.py <net> <B>
Per sweep launch of one forward and per phase, over the workgroups (us after the launch's first stamp): when the phase was entered, when the last gather
wave was done, when the last chain wave was done -- i.e. the drain (chain waves alone) and the ramp (barrier -> weights staged) that remain INSIDE a
sweep, next to the spread between workgroups that a kernel boundary would turn into waiting."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(30):
    m.forward_device(*args)
torch.cuda.synchronize()
n = 4 * 256 * 8 * 16 * 4
buf = (C.c_ulonglong * n)()
lib.gnnb_debug_sweep_wall.argtypes = [C.c_void_p, C.c_int]
assert lib.gnnb_debug_sweep_wall(buf, n) == 0
w = np.array(buf[:], dtype=np.uint64).astype(np.int64).reshape(4, 256, 8, 16, 4)


def pct(x):
    return " / ".join(f"{np.percentile(x, q):6.1f}" for q in (0, 50, 100))


print(f"{net} B={B}: k_sweep stamps of the last forward (us; min / median / max over the 256 workgroups)")
for o in range(4):
    ww = w[o]
    if not ww[:, 0, :, 0].any():
        continue
    nph = max(p + 1 for p in range(8) if ww[:, p, :, 0].any())
    t0 = ww[:, 0, :, 0][ww[:, 0, :, 0] > 0].min()
    print(f" sweep launch {o}: {nph} phase(s)")
    prev_end = None
    for p in range(nph):
        ent = (ww[:, p, :, 0].max(1) - t0) * 0.01                 # last wave through the barrier
        g_end = (ww[:, p, :8, 3].max(1) - t0) * 0.01
        c_stg = (ww[:, p, 8:, 1].max(1) - t0) * 0.01
        c_end = (ww[:, p, 8:, 3].max(1) - t0) * 0.01
        print(f"  phase {p}: entered {pct(ent)} | weights staged +{pct(c_stg - ent)} | last gather wave done {pct(g_end)} | last chain wave done {pct(c_end)} | "
              f"drain (chains after the last gather) {pct(c_end - g_end)} | phase length {pct(c_end - ent)}")
    print(f"  workgroups done {pct(c_end)}: a kernel boundary after every phase would cost each its max - median; inside the sweep only the last one does")
