#!/usr/bin/env python
"""dev helper (GPU box, under rocprofv3 --pmc ... --kernel-trace): a few forwards of cifar_base_kw B=256 without checking the results
(TOP_STOP builds leave k_top early: wrong results, timing / counters only).  tools/run.sh top_phases drives it."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state
m = GraphNet(2, 64)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in shipped_state().items()})
net = sys.argv[1] if len(sys.argv) > 1 else "cifar_base_kw"
batch = synth.make_batch(net, int(sys.argv[2]) if len(sys.argv) > 2 else 256, seed=1234)
dev = torch.device("cuda")
args = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in batch.forward_args()]
args[4] = batch.primal_inputs.to(dev); args[6] = batch.masks.to(dev)
for _ in range(6):
    m.forward_device(*args)
torch.cuda.synchronize()
