"""Branch-and-bound on one robustness property with the MI355X scorer and the Gurobi-free LP producer.

    python examples/bab_demo.py [--net cifar_base_kw] [--eps 0.03] [--nodes 40] [--babsr | --threshold 0.2]

--threshold T runs the reference loop's own control flow (relu_conv_gnnkwthreshold.py:150-199): a GNN decision whose improvement of the bound is
below T makes the loop ask the BaBSR heuristic too (on the device), bound its children and keep the better pair; try --eps 0.09.

Prints the trace of plnn/relu_conv_gnnkwthreshold.py:202 for every branch.  Needs the GPU library (no CPU fallback)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gnn_branching_amd import lp_producer, nets                     # noqa: E402
from gnn_branching_amd.graphnet.graph_score import GraphChoice      # noqa: E402

CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models", "cifar_trained_gnn",
                    "best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="cifar_base_kw")
    ap.add_argument("--eps", type=float, default=0.03)
    ap.add_argument("--nodes", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--babsr", action="store_true", help="branch with the BaBSR heuristic instead of the GNN")
    ap.add_argument("--threshold", type=float, default=None, help="branching_threshold of the GNN + KW fall-back loop (the reference uses 0.2)")
    args = ap.parse_args()
    layers = nets.load_verified_net(args.net, 3, 5)
    x = torch.from_numpy(np.random.RandomState(args.seed).standard_normal((3, 32, 32)).astype(np.float32))
    lp = lp_producer.LayerGraphLP(layers, x - args.eps, x + args.eps)
    root_mask = [torch.full((int(np.prod(lp.shapes[i + 1])),), -1, dtype=torch.long) for i in lp.pre_relu_indices]
    root = lp.solve(root_mask)
    print(f"root: lb {root.lb:.5f} ub {root.ub:.5f}, undecided ReLUs per layer {[int((m == -1).sum()) for m in root.mask]}")
    if args.threshold is not None:
        from gnn_branching_amd.plnn.kw_score_conv import choose_node_conv
        choice = GraphChoice(root.mask, CKPT)

        def kw(sub, icp, random_order, sparsest_layer):
            return choose_node_conv(sub.lower_all, sub.upper_all, sub.mask, lp.layers, lp.pre_relu_indices, icp, random_order, sparsest_layer)
        glb, gub, solves, branches, n_kw, n_used = lp_producer.branch_and_bound_threshold(
            lp, lp_producer.gnn_scorer(choice, lp), kw, layers, max_branches=args.nodes // 2, decision_bound=0.0, branching_threshold=args.threshold)
        verdict = "property holds" if glb >= 0 else ("counter-example found" if gub < 0 else "undecided within the node budget")
        print(f"after {branches} branches ({solves} LP solves; {n_kw} bounded a KW decision, {n_used} kept it): lb {glb:.5f} ub {gub:.5f} -> {verdict}")
        return
    if args.babsr:
        scorer = lp_producer.babsr_scorer(lp)
    else:
        choice = GraphChoice(root.mask, CKPT)
        scorer = lp_producer.gnn_scorer(choice, lp)
    glb, gub, visited = lp_producer.branch_and_bound(lp, scorer, layers, max_nodes=args.nodes, decision_bound=0.0)
    verdict = "property holds" if glb >= 0 else ("counter-example found" if gub < 0 else "undecided within the node budget")
    print(f"after {visited} LP solves: lb {glb:.5f} ub {gub:.5f} -> {verdict}")


if __name__ == "__main__":
    main()
