#!/usr/bin/env python
"""dev helper (GPU box): where the waves of ONE k_gather_update_q template spend their cycles (library built with -DFUSED_TIMING=6
[-DQ_TIME_LANES=.. -DQ_TIME_SRC=.. -DQ_TIME_POST=..], GNNB_LIB=...).   python tools/fusedq_timing.py <net> <B>"""
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
# one more forward alone: start / end of every wave of the instrumented launch on the chip-wide 100 MHz clock, beside its HIP-event duration
eng = m.engine()
eng.profile_enable(True)
eng.profile_read(reset=True)
eng.profile_trace(65536)
m.forward_device(*args)
torch.cuda.synchronize()
eng.profile_read(reset=True)
trace = eng.profile_trace(65536)
eng.profile_enable(False)
wall = (C.c_ulonglong * (2 * 16 * 256))()
lib.gnnb_debug_wall(wall, 2 * 16 * 256)
w = np.array(wall[:], dtype=np.uint64).reshape(-1, 2).astype(np.int64)
wg_end = (w[:, 1].reshape(-1, 16).max(1) - w[:, 0].min()) * 0.01          # (rows are written per workgroup: 16 waves each)
g_end = (w[:, 1].reshape(-1, 16)[:, :8].max(1) - w[:, 0].min()) * 0.01
print("per workgroup, us after the first wave's start: last chain wave ends min / 10 % / median / 90 % / max = " +
      " / ".join(f"{np.percentile(wg_end, q):.1f}" for q in (0, 10, 50, 90, 100)) + "; last gather wave ends " +
      " / ".join(f"{np.percentile(g_end, q):.1f}" for q in (0, 10, 50, 90, 100)))
print("median / max end of the last chain wave by XCD (blockIdx % 8): " + "  ".join(f"{x}: {np.median(wg_end[x::8]):.1f} / {wg_end[x::8].max():.1f}" for x in range(8)))
print(f"instrumented launch: first wave start -> last wave end {(w[:, 1].max() - w[:, 0].min()) * 0.01:.1f} us; starts spread over {(w[:, 0].max() - w[:, 0].min()) * 0.01:.1f} us, "
      f"ends over {(w[:, 1].max() - w[:, 1].min()) * 0.01:.1f} us; HIP-event durations of this forward's launches: {[(nm, round(1e3 * ms, 1)) for nm, ms in trace]}")
names = ["G tables staged, header", "G tile decode + bounds", "G table build + walk", "G wait for ring slot", "G rows -> ring",
         "C tables staged, header", "C weights staged (all)", "C wait for a tile", "C rows->regs, chain, stores", "-"]
ng, nc = out[15], out[14]
print(f"{net} B={B}: {ng / n:.0f} gather waves, {nc / n:.0f} chain waves per forward (cycles per wave, shader clock)")
for i, nm in enumerate(names):
    d = ng if i < 5 else nc
    print(f"  {nm:30s} {out[i] / max(d, 1):9.0f}")
print(f"  gather wave total {sum(out[:5]) / max(ng, 1):9.0f}   chain wave total {sum(out[5:10]) / max(nc, 1):9.0f}")
