"""Randomized parity sweep: the HIP scorer against the CPU oracle over batch sizes (incl. odd ones), seeds, both weight sets
and mask patterns -- the BaB mask as generated, everything undecided (dead nodes scored too), a sparse subset with one sample
that has nothing to score -- each time with the whole workspace poisoned with NaN before the call."""
import numpy as np
import pytest
import torch

from gnn_branching_amd import synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import SCORE_ATOL, score_tol, state_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("net", ["cifar_base_kw", "cifar_wide_kw", "cifar_deep_kw"])
def test_random_batches_masks_and_weights(net):
    from oracle import gnn_oracle
    torch.set_num_threads(min(16, torch.get_num_threads()))
    models = {}
    for fam in ("shipped", "random"):
        m = GraphNet(2, 64)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in state_of(fam).items()})
        models[fam] = m
    worst, n = {"shipped": 0.0, "random": 0.0}, 0
    for B in (1, 3, 8, 17):
        for seed in (100, 101, 102):
            batch = synth.make_batch(net, B, seed=seed + B)
            args = list(batch.forward_args())
            rng = np.random.RandomState(seed)
            if seed % 3 == 1:
                args[6] = torch.ones_like(batch.masks)
            elif seed % 3 == 2:
                args[6] = batch.masks * torch.from_numpy((rng.uniform(size=tuple(batch.masks.shape)) < 0.3).astype(np.float32))
                args[6][0] = 0
            for fam, model in models.items():
                with torch.no_grad():
                    want = gnn_oracle.padded_scores(gnn_oracle.oracle_forward(state_of(fam), *args), args[6]).numpy()
                    model.forward_device(*args)
                    model.engine().workspace(B).view(torch.float32).fill_(float("nan"))
                    res = model.forward_device(*args).check()
                got = res.scores.cpu().numpy()
                fin = np.isfinite(want)
                assert np.array_equal(np.isfinite(got), fin), (net, B, seed, fam)
                if fin.any():
                    err = float(np.abs(got[fin] - want[fin]).max())
                    worst[fam] = max(worst[fam], err)
                    assert err <= score_tol(fam, want[fin]), (net, B, seed, fam, err)
                dec = res.decisions.cpu().tolist()
                for b in range(B):
                    if not fin[b].any():
                        assert dec[b] == [-1, -1]
                n += 1
    print(f"{net}: {n} cases, worst |score - oracle| shipped {worst['shipped']:.3e}, random {worst['random']:.3e}")
