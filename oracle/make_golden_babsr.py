#!/usr/bin/env python
"""Golden vectors for the BaBSR scorer from the REFERENCE ITSELF (authoring container only; TEST INFRASTRUCTURE).

Imports /root/reference/plnn/kw_score_conv.py unmodified (it has no Gurobi dependency) and runs ``choose_node_conv``
(:41-156, gt=True) on every subproblem of the committed golden batches (tests/golden/<net>_B*.npz), under several
(sparsest_layer, icp_score_counter, decision_threshold) settings so that the three branches of the decision rule are
exercised.  Stores scores and decisions in tests/golden/<case>_babsr.npz.

Usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/make_golden_babsr.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REF = os.environ.get("GNNB_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.abspath(REPO))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import plnn.kw_score_conv as ref_kw                      # noqa: E402  (the reference)
from plnn.modules import Flatten as RefFlatten           # noqa: E402
from gnn_branching_amd.plnn.modules import Flatten as OurFlatten   # noqa: E402
from tests.common import GOLDEN_CASES, load_golden       # noqa: E402

SETTINGS = [  # (sparsest_layer, icp_score_counter, decision_threshold)
    (0, 0, 0.001),        # the call made by relu_gnn (relu_conv_gnnkwthreshold.py:157)
    (-1, 0, 0.001),
    (1, 1, 0.001),
    (0, 0, 1e9),          # scores never informative -> intercept branch
    (0, 2, 1e9),          # ... and intercept budget used up -> preferred-layer branch
]


def main():
    for case in GOLDEN_CASES:
        g, batch = load_golden(case)
        B = batch.batch_size
        fixed = [RefFlatten() if isinstance(l, OurFlatten) else l for l in batch.layers["fixed_layers"]]
        nlay = len(fixed) + 1
        pre_relu = [i for i, l in enumerate(fixed) if isinstance(l, torch.nn.ReLU)]
        rec = {"settings": np.array(SETTINGS, dtype=np.float64)}
        L = len(pre_relu)
        for b in range(B):
            one = batch.slice(b, b + 1)
            layers = fixed + [one.layers["prop_layers"][0]]
            # per-layer bounds list as the BaB loop holds it: only the pre-ReLU entries are read (:82)
            lbs, ubs = [None] * (nlay + 1), [None] * (nlay + 1)
            for k, i in enumerate(pre_relu):
                lbs[i], ubs[i] = one.lower_bounds_all[k + 1][0], one.upper_bounds_all[k + 1][0]
            for i, l in enumerate(fixed):                 # Flatten reshapes with lower_bounds[layer_idx].size() (:115)
                if isinstance(l, RefFlatten):
                    lbs[i] = one.lower_bounds_all[[j for j, q in enumerate(pre_relu) if q < i][-1] + 1][0]
            mask = [m[0] for m in one.bab_masks]
            for si, (sp, icp, thr) in enumerate(SETTINGS):
                random_order = list(range(L))
                dec, cnt, score = ref_kw.choose_node_conv(lbs, ubs, mask, layers, pre_relu, icp, random_order, sp,
                                                          decision_threshold=thr, gt=True)
                rec[f"dec_{b}_{si}"] = np.array(dec + [cnt], np.int64)
                if si == 0:
                    rec[f"score_{b}"] = torch.cat(score).numpy()
        np.savez_compressed(os.path.join(REPO, "tests", "golden", case + "_babsr.npz"), **rec)
        print(case, "decisions", [rec[f"dec_{b}_0"].tolist() for b in range(B)],
              "branches", [[rec[f"dec_{b}_{si}"].tolist() for si in range(len(SETTINGS))] for b in range(1)])


if __name__ == "__main__":
    main()
