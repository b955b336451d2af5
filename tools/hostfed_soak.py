#!/usr/bin/env python
"""dev helper (GPU box): soak of engine.HostFedPipeline with the record image (dual_vars / primals as records of the ambiguous nodes) -- N submits per
configuration cycling THREE batches with different ambiguous sets (LP-like signed duals in one of them) through two slots, pinned and pageable inputs;
every result compared bit for bit with the device-resident forward and every status word checked.   python3 tools/hostfed_soak.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.engine import HostFedPipeline
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import shipped_state

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda")
for net, B, pinned in (("cifar_base_kw", 256, True), ("cifar_deep_kw", 128, True), ("cifar_wide_kw", 64, False), ("cifar_base_kw", 3, True), ("cifar_deep_kw", 31, False)):
    m = GraphNet(2, 64)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in shipped_state().items()})
    eng = m.engine()
    sets, want = [], []
    for j, seed in enumerate((31, 32, 33)):
        batch = synth.make_batch(net, B, seed=seed)
        a = list(batch.forward_args())
        if j == 1:      # duals of either sign, as an LP produces them
            rng = np.random.RandomState(7)
            a[2] = [torch.from_numpy((rng.standard_normal(tuple(t.shape)) * 0.05).astype(np.float32)) for t in a[2]]
        d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in a]
        d[4], d[6] = a[4].to(dev), a[6].to(dev)
        with torch.no_grad():
            r = eng.forward(*d).check()
        want.append((r.scores.clone(), r.decisions.clone()))
        h = [[t.float().contiguous() for t in g] if isinstance(g, list) else g for g in a]
        h[4], h[6] = a[4].float().contiguous(), a[6].float().contiguous()
        if pinned:
            h = [[t.pin_memory() for t in g] if isinstance(g, list) else g for g in h]
            h[4], h[6] = h[4].pin_memory(), h[6].pin_memory()
        sets.append(h)
    pipe = HostFedPipeline(eng)
    t0 = time.time()
    pending = []
    with torch.no_grad():
        for i in range(N):
            k = (i * 7 + i // 5) % 3
            pending.append((k, pipe.submit(*sets[k])))
            if len(pending) >= 6:                       # results held across more than `depth` submits
                kk, r = pending.pop(0)
                r.check()
                assert torch.equal(r.scores, want[kk][0]) and torch.equal(r.decisions, want[kk][1]), (net, B, i)
    for kk, r in pending:
        r.check()
        assert torch.equal(r.scores, want[kk][0]) and torch.equal(r.decisions, want[kk][1]), (net, B)
    print(f"{net} B={B} ({'pinned' if pinned else 'pageable'}): {N} submits of three batches through two slots identical to the device-resident forward, "
          f"status clean, {pipe.link_bytes} bytes over the link per batch ({time.time() - t0:.1f}s)", flush=True)
