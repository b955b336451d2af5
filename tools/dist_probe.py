#!/usr/bin/env python
"""dev helper (GPU box): does an initialised RCCL process group slow the forward down?  Run as
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 tools/dist_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state
import torch.distributed as dist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
m = GraphNet(2, 64)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in shipped_state().items()})
eng = m.engine()
batch = synth.make_batch("cifar_base_kw", 256, seed=1234)
args = batch.forward_args()
d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
d[4], d[6] = args[4].to(dev), args[6].to(dev)


def t(label):
    for _ in range(20):
        eng.forward(*d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.forward(*d)
    torch.cuda.synchronize()
    print(f"{label:50s} {1e3 * (time.perf_counter() - t0) / 50:.4f} ms per forward", flush=True)


t("before init_process_group")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t("after init_process_group (no collective yet)")
x = torch.ones(1 << 20, device=dev)
out = torch.empty(1 << 20, device=dev)
dist.all_gather_into_tensor(out, x)
torch.cuda.synchronize()
t("after the first all-gather (communicator exists)")
dist.barrier()
torch.cuda.synchronize()
t("after a barrier")
dist.destroy_process_group()
t("after destroy_process_group")
