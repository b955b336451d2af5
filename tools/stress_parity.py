#!/usr/bin/env python
"""dev helper (GPU box): randomized parity sweep -- HIP scorer vs the CPU oracle over seeds, batch sizes, networks, weight
sets and mask patterns (incl. everything undecided / sparse masks), each with the workspace poisoned with NaN first."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from oracle import gnn_oracle
from tests.common import shipped_state

torch.set_num_threads(16)
worst = 0.0
n = 0
t0 = time.time()
states = {"shipped": shipped_state(), "random": gnn_oracle.random_gnn_state(77)}
models = {}
for fam, st in states.items():
    m = GraphNet(2, 64); m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()}); models[fam] = m
for net in ("cifar_base_kw", "cifar_wide_kw", "cifar_deep_kw"):
    for B in (1, 2, 3, 5, 8, 17):
        for seed in (100, 101, 102):
            batch = synth.make_batch(net, B, seed=seed + B)
            args = list(batch.forward_args())
            rng = np.random.RandomState(seed)
            mode = seed % 3
            if mode == 1:                                   # everything undecided (dead nodes scored too)
                args[6] = torch.ones_like(batch.masks)
            elif mode == 2:                                 # a sparse subset of the ambiguous nodes, one sample with none
                keep = torch.from_numpy((rng.uniform(size=tuple(batch.masks.shape)) < 0.3).astype(np.float32))
                args[6] = batch.masks * keep
                args[6][0] = 0
            for fam in ("shipped", "random"):
                model = models[fam]
                with torch.no_grad():
                    want = gnn_oracle.padded_scores(gnn_oracle.oracle_forward(states[fam], *args), args[6]).numpy()
                    model.forward_device(*args)
                    model.engine().workspace(B).view(torch.float32).fill_(float("nan"))
                    res = model.forward_device(*args).check()
                got = res.scores.cpu().numpy()
                fin = np.isfinite(want)
                assert np.array_equal(np.isfinite(got), fin), (net, B, seed, fam)
                err = float(np.abs(got[fin] - want[fin]).max()) if fin.any() else 0.0
                worst = max(worst, err)
                n += 1
                assert err <= 1e-4, (net, B, seed, fam, err)
                dec = res.decisions.cpu().tolist()
                for b in range(B):
                    row = want[b]
                    if not np.isfinite(row).any():
                        assert dec[b] == [-1, -1]
    print(net, "ok so far:", n, "cases, worst |score - oracle| =", worst, f"({time.time() - t0:.0f}s)", flush=True)
print("all", n, "cases passed; worst error", worst)
