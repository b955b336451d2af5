"""``graphnet.graph_score`` surface of the reference: ``GraphChoice``.

Same constructor and ``decision`` signature as reference graph_score.py:6-56 so
``plnn/relu_conv_gnnkwthreshold.py`` (call sites :109, :117, :230, :239) can use it
unchanged.  ``decision_batch`` is the batched entry point the MI355X build adds.
"""
import sys
import time

import numpy as np
import torch

from .graph_conv import GraphNet


def _load_state(model_name):
    if str(model_name).endswith(".npz"):
        d = dict(np.load(model_name))
        order = [str(k) for k in d.pop("__order__")] if "__order__" in d else list(d)
        return {k: torch.from_numpy(d[k]) for k in order}
    # the shipped checkpoint is a legacy pickle with CUDA-tagged storages (graph_score.py:11 passes no map_location)
    return torch.load(model_name, map_location="cpu", weights_only=True)


class GraphChoice:

    def __init__(self, init_mask, model_name, linear=False):
        model = GraphNet(2, 64)                                   # graph_score.py:9
        model.load_state_dict(_load_state(model_name))            # :11
        model.eval()
        self.model = model                                        # device copies live in the HIP engine
        trans_len, temp = [], 0
        for i in init_mask:                                       # :14-19
            temp += len(i)
            trans_len.append(temp)
        self.trans_len = torch.tensor(trans_len)
        self.verbose = True

    @staticmethod
    def _mask_1d(mask):
        return torch.cat([(i == -1).float().reshape(-1) for i in mask], 0)      # :22-23

    def decision(self, lower_bounds_all, upper_bounds_all, dual_vars, primal_input, primals, layers, mask):
        """[dec_lay, dec_idx] of the highest-scoring undecided ReLU (NOTE the argument order
        ``primal_input, primals`` -- reference graph_score.py:21)."""
        mask_1d = self._mask_1d(mask).unsqueeze(0)
        start = time.time()
        host = not any(torch.is_tensor(t) and t.is_cuda for grp in (lower_bounds_all, upper_bounds_all, dual_vars, primals) for t in grp) \
            and not (torch.is_tensor(primal_input) and primal_input.is_cuda)
        with torch.no_grad():
            if host:                                              # the reference's pattern: CPU tensors -> one pinned transfer
                d, _ = self.model.engine().forward_host(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_input,
                                                        layers, mask_1d)
                dec = d[0].tolist()
            else:
                res = self.model.forward_device(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_input,
                                                layers, mask_1d)
                res.check()
                dec = res.decisions[0].tolist()
        end = time.time()
        if self.verbose:
            print(f'graph requires: {end-start}')                 # :36
        if dec[0] < 0:
            # reference: torch.max over an empty score tensor raises (graph_score.py:41)
            print("[gnn_branching_amd] GraphChoice.decision: no undecided ReLU in the mask", file=sys.stderr)
            raise RuntimeError("GraphChoice.decision: no undecided ReLU in the mask")
        return [int(dec[0]), int(dec[1])]

    def decision_batch(self, lower_bounds_all, upper_bounds_all, dual_vars, primal_input, primals, layers, masks_1d):
        """Batched decisions: inputs carry a leading batch dimension B (same layout as
        GraphNet.forward), ``masks_1d`` is (B, R) with 1 where the BaB mask is -1.
        Returns the device-resident result; ``.decisions`` is a (B, 2) int32 tensor of [layer, idx]."""
        with torch.no_grad():
            return self.model.forward_device(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_input,
                                             layers, masks_1d)
