#!/usr/bin/env python
"""dev helper (GPU box): soak of the default path -- N forwards per configuration, every result compared bit for bit with the
first one and the status word checked each time (the LDS queues of the fused half-passes wrap ~10^5 times per forward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
# (round 3: the small batches cover k_top's four-workgroup hand-offs, k_scored_tail and k_classify_pre)
for net, B in (("cifar_base_kw", 256), ("cifar_wide_kw", 256), ("cifar_deep_kw", 128), ("cifar_base_kw", 7), ("cifar_deep_kw", 1),
               ("cifar_base_kw", 1), ("cifar_base_kw", 2), ("cifar_wide_kw", 3), ("cifar_base_kw", 33), ("cifar_deep_kw", 64), ("cifar_base_kw", 100), ("cifar_wide_kw", 128)):
    m = GraphNet(2, 64)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in shipped_state().items()})
    batch = synth.make_batch(net, B, seed=11)
    dev = torch.device("cuda")
    args = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in batch.forward_args()]
    args[4] = batch.primal_inputs.to(dev); args[6] = batch.masks.to(dev)
    t0 = time.time()
    with torch.no_grad():
        first = m.forward_device(*args).check()
        ref_s, ref_d = first.scores.clone(), first.decisions.clone()
        for i in range(N):
            r = m.forward_device(*args)
            if i % 10 == 9 or i == N - 1 or B <= 64:
                r.check()
                assert torch.equal(r.scores, ref_s) and torch.equal(r.decisions, ref_d), (net, B, i)
    print(f"{net} B={B}: {N} forwards identical, status clean ({time.time() - t0:.1f}s)", flush=True)
