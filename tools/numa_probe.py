#!/usr/bin/env python
"""dev helper (GPU box): H2D rate and overlap with the forward for pinned buffers allocated by a thread on the GPU's NUMA node vs the other one."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
print("GPU", bdf, "numa node", node, "local cpus", open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip())


def cpus_of(n):
    out = set()
    for part in open(f"/sys/devices/system/node/node{n}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out |= set(range(int(a), int(b or a) + 1))
    return out


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
dst = torch.empty(18 * P, dtype=torch.float32, device=dev)
s = torch.cuda.Stream(device=dev)
K = 40
allowed = os.sched_getaffinity(0)


def timeit(f):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


for alloc_node in (node, 1 - node):
    for run_node in (node, 1 - node):
        os.sched_setaffinity(0, cpus_of(alloc_node) & allowed)
        pin = torch.empty(18 * P, dtype=torch.float32, pin_memory=True)
        pin.fill_(1.0)
        os.sched_setaffinity(0, cpus_of(run_node) & allowed)

        def copies():
            with torch.cuda.stream(s):
                for j in range(18):
                    dst[j * P:(j + 1) * P].copy_(pin[j * P:(j + 1) * P], non_blocking=True)

        def both():
            copies()
            eng.forward(*d)
        tc = timeit(copies)
        print("pinned on node %d (%s), thread on node %d (%s): copies alone %.3f ms = %.1f GB/s; copies + forward %.3f ms; forward alone %.3f ms" % (
            alloc_node, "GPU's" if alloc_node == node else "other", run_node, "GPU's" if run_node == node else "other", tc, 18 * P * 4 / tc / 1e6, timeit(both),
            timeit(lambda: eng.forward(*d))), flush=True)
        del pin
os.sched_setaffinity(0, allowed)
