#!/usr/bin/env python
"""Convert the reference's DATA files (trained weights) into .npz assets.

TEST/BUILD INFRASTRUCTURE -- runs only in the authoring container, where
/root/reference is mounted.  Nothing here is imported by the product path.

Inputs (data, MIT-licensed, (c) 2019 oval-group -- see NOTICE in assets/):
  models/cifar_trained_gnn/best_snapshot_..._epoch_57.pt   trained GNN, 52 fp32 tensors
        (path hard-wired at reference experiments/bab_mip.py:35-36)
  models/cifar_{base,wide,deep}_kw.pth   verified networks, d['state_dict'][0]
        (loaded at reference exp_utils/model_utils.py:214-225)

Outputs: gnn_branching_amd/assets/*.npz  (plain numpy arrays, key order kept).
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GNNB_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "gnn_branching_amd", "assets")

GNN_PT = "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt"


def main():
    os.makedirs(OUT, exist_ok=True)
    sd = torch.load(os.path.join(REF, GNN_PT), map_location="cpu", weights_only=True)
    arrs = {k: v.numpy() for k, v in sd.items()}
    # keep the state-dict order explicitly: npz does not promise key order
    np.savez(os.path.join(OUT, "cifar_trained_gnn.npz"),
             __order__=np.array(list(arrs.keys())), **arrs)
    print("gnn:", len(arrs), "tensors", sum(a.size for a in arrs.values()), "params")
    for name in ("base", "wide", "deep"):
        d = torch.load(os.path.join(REF, f"models/cifar_{name}_kw.pth"),
                       map_location="cpu", weights_only=False)
        sd = d["state_dict"][0]
        arrs = {k: v.numpy() for k, v in sd.items()}
        np.savez(os.path.join(OUT, f"cifar_{name}_kw.npz"),
                 __order__=np.array(list(arrs.keys())), **arrs)
        print(name, {k: a.shape for k, a in arrs.items()})


if __name__ == "__main__":
    sys.exit(main())
