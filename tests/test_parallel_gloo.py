"""The N>1 path on CPU: world_size-2/3 gloo ranks shard a batch, score their shard (the oracle stands in for the
GPU scorer here) and all-gather the padded scores; the result must equal the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gnn_branching_amd import parallel


def test_shard_bounds_cover_the_batch():
    for B in (1, 2, 3, 7, 256, 1000):
        for W in (1, 2, 3, 4, 8):
            spans = [parallel.shard_bounds(B, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = parallel.shard_sizes(B, W)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == B


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gnn_oracle
        from tests.common import load_golden, relu_sizes, state_of
        g, batch = load_golden(case)

        def score_fn(shard):
            with torch.no_grad():
                s = gnn_oracle.oracle_forward(state_of("random"), *shard.forward_args())
            return gnn_oracle.padded_scores(s, shard.masks)
        full = parallel.score_sharded(batch, score_fn)
        dec = parallel.decisions_from_scores(full, relu_sizes(batch))
        # the pipelined form bench.py uses (gather of batch i not waited for until batch i + 1 has been launched): same bytes
        lo, hi = parallel.shard_bounds(batch.batch_size, world, rank)
        local = score_fn(batch.slice(lo, hi))
        h1 = parallel.gather_scores_async(local, batch.batch_size)
        h2 = parallel.gather_scores_async(local + 1.0, batch.batch_size)        # a second collective in flight behind it
        assert torch.equal(h1.wait(), full) and torch.equal(h2.wait(), full + 1.0)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=full.numpy(), dec=dec.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scoring_equals_single_process(tmp_path, world):
    case = "cifar_base_kw_B3"          # 3 subproblems: shards 2+1 (ragged) and 1+1+1
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    from oracle import gnn_oracle
    from tests.common import load_golden, state_of
    g, batch = load_golden(case)
    with torch.no_grad():
        one = gnn_oracle.oracle_forward(state_of("random"), *batch.forward_args())
    want = gnn_oracle.padded_scores(one, batch.masks).numpy()
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        # per-sample results do not depend on what else is in the shard up to the oracle's own batching noise (~4e-6)
        fin = np.isfinite(want)
        assert np.array_equal(np.isinf(got["scores"]), np.isinf(want))
        assert np.abs(got["scores"][fin] - want[fin]).max() <= 1e-5
        assert got["dec"].tolist() == g["random_decisions"].tolist()
    # every rank ends up with the same bytes
    a = np.load(os.path.join(tmp_path, "rank0.npz"))["scores"]
    for r in range(1, world):
        assert np.array_equal(a, np.load(os.path.join(tmp_path, f"rank{r}.npz"))["scores"])
