#!/usr/bin/env python
"""Generate tests/golden/*_online.npz from the REFERENCE's own online-learning GraphChoice (authoring container only).

TEST INFRASTRUCTURE.  Imports /root/reference/graphnet/graph_score_online.py unmodified with the host-side shims of
make_golden.py (``.cuda()`` -> no-op, ``torch.load`` -> map_location='cpu').  For one synthetic subproblem
(gnn_branching_amd/synth.py; the inputs are regenerated from the seed by the tests, not stored) it runs, twice in a row on
the same GraphChoice (so the second step sees the Adam state of the first):

    d = g.decision(...)   ;   g.online_learning(kw_decision, improvement)

with a KW decision different from the GNN's, and stores the decisions, the loss, the gradient of every parameter after
each backward and the parameters after each optimizer step.

Usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/make_golden_online.py
"""
import os
import sys

import numpy as np
import torch
from torch import nn

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REF = os.environ.get("GNNB_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.abspath(REPO))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self
_orig_load = torch.load
torch.load = lambda f, *a, **k: _orig_load(f, map_location="cpu", weights_only=True)

import graphnet.graph_score_online as ref_online    # noqa: E402  (the reference)
from plnn.modules import Flatten as RefFlatten      # noqa: E402

from gnn_branching_amd import synth                 # noqa: E402
from gnn_branching_amd.plnn.modules import Flatten as OurFlatten  # noqa: E402
from oracle.gnn_oracle import random_gnn_state      # noqa: E402

GNN_PT = os.path.join(REF, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
OUT = os.path.join(REPO, "tests", "golden")
RANDOM_SEED = 20240917
LR, WD = 1e-4, 1e-4
CASES = [("cifar_base_kw", 3, 0, [(3, 5), (3, 5), (7, 1)], 0)]      # (net, B of the synthetic batch, seed, props, sample used)
IMPROVEMENTS = [0.125, 0.03]


def ref_layers(layers):
    return {"fixed_layers": [RefFlatten() if isinstance(l, OurFlatten) else l for l in layers["fixed_layers"]],
            "prop_layers": layers["prop_layers"]}


def kw_choice(one, gnn_decision, step):
    """A deterministic stand-in for the KW decision: the (step+1)-th undecided ReLU of the LAST layer that has one and
    is not the GNN's decision, as [layer, index]."""
    for lay in reversed(range(len(one.bab_masks))):
        idx = (one.bab_masks[lay][0].reshape(-1) == -1).nonzero().view(-1).tolist()
        idx = [i for i in idx if [lay, i] != list(gnn_decision)]
        if len(idx) > step:
            return [lay, idx[step]]
    raise RuntimeError("no KW candidate")


def main():
    rnd = random_gnn_state(RANDOM_SEED)
    for net, B, seed, props, sample in CASES:
        batch = synth.make_batch(net, B, seed=seed, props=props)
        one = batch.slice(sample, sample + 1)
        init_mask = [m[0] for m in one.bab_masks]
        rec = {"net": np.array(net), "B": np.array(B), "seed": np.array(seed), "props": np.array(props), "sample": np.array(sample),
               "random_seed": np.array(RANDOM_SEED), "lr": np.array(LR), "wd": np.array(WD), "improvements": np.array(IMPROVEMENTS)}
        for fam, src in (("shipped", GNN_PT), ("random", rnd)):
            if isinstance(src, str):
                g = ref_online.GraphChoice(init_mask, src, lr=LR, wd=WD)
            else:
                tmp = "/tmp/_gnnb_golden_online_state.pt"
                torch.save({k: torch.as_tensor(np.asarray(v)) for k, v in src.items()}, tmp)
                g = ref_online.GraphChoice(init_mask, tmp, lr=LR, wd=WD)
                os.remove(tmp)
            for step, imp in enumerate(IMPROVEMENTS):
                d = g.decision(one.lower_bounds_all, one.upper_bounds_all, one.dual_vars, one.primal_inputs,
                               [p.tolist() for p in one.primals], ref_layers(one.layers), init_mask)
                kw = kw_choice(one, d, step)
                loss = float(g.gnn_score - g.scores[0][len(g.mask_1d[0][:(0 if kw[0] == 0 else int(g.trans_len[kw[0] - 1])) + kw[1]].nonzero())] + imp)
                g.online_learning(kw, imp)
                params = list(g.model.parameters())
                rec[f"{fam}_s{step}_decision"] = np.array(d, np.int32)
                rec[f"{fam}_s{step}_kw"] = np.array(kw, np.int32)
                rec[f"{fam}_s{step}_loss"] = np.array(loss, np.float32)
                rec[f"{fam}_s{step}_grad"] = np.concatenate([p.grad.detach().numpy().reshape(-1) for p in params]).astype(np.float32)
                rec[f"{fam}_s{step}_params"] = np.concatenate([p.detach().numpy().reshape(-1) for p in params]).astype(np.float32)
                g.del_score()
                gr = rec[f"{fam}_s{step}_grad"]
                print(net, fam, "step", step, "decision", d, "kw", kw, "loss", loss, "|grad| max", float(np.abs(gr).max()),
                      "nonzero", int((gr != 0).sum()))
        path = os.path.join(OUT, f"{net}_online.npz")
        np.savez_compressed(path, **rec)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
