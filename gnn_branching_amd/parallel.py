"""Data-parallel scoring of a batch of live BaB subproblems across GPUs.

Subproblems are independent (no cross-sample term anywhere in graph_conv.py:77-470), so the batch is cut into
contiguous shards, one per rank (one process per GPU), every rank scores its shard with the same replicated GNN and
verified-network weights, and ONE all-gather of the padded score matrix (RCCL over xGMI with backend "nccl", gloo on
CPU for the tests) hands every rank the scores of the whole batch for the branch selector.  There is no reduction, so
the N-rank result equals the 1-rank result bit for bit.

The scorer itself is passed in (``score_fn(shard) -> (B_local, R) tensor``): on the GPU it is
``GraphNet.forward_device(...).scores``; the CPU tests plug in a CPU stand-in to exercise exactly this sharding logic.
"""
import torch
import torch.distributed as dist

# Handle options (include/gnnb.h gnnb_set_option) of every scorer that runs while a collective may be in flight on the same GPU: k_top must
# not spread a sample over workgroups that spin-wait on each other -- they need all their partners resident at one per CU, and an RCCL
# kernel on the communication stream holds CUs for as long as its peers take.  bench.py's multi-GPU path and engine.BatchPipeline (which
# sets the same option itself) create their handles with it; S = 1 computes the same bits (tests/test_gpu_parity.py, test_gpu_dist_safety.py).
DIST_ENGINE_OPTIONS = {"top_split": 1}


def shard_bounds(batch_size, world_size, rank):
    """Contiguous shard [lo, hi) of rank ``rank``: sizes differ by at most one, earlier ranks take the remainder."""
    base, rem = divmod(batch_size, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(batch_size, world_size):
    return [shard_bounds(batch_size, world_size, r)[1] - shard_bounds(batch_size, world_size, r)[0] for r in range(world_size)]


def gather_scores(local_scores, batch_size, group=None):
    """All-gather the per-rank (B_r, R) padded scores into the (batch_size, R) matrix of the whole batch.

    Equal shards use one ``all_gather_into_tensor`` (a single RCCL collective); ragged shards are padded to the
    largest shard so it still is ONE collective, and the padding rows are dropped afterwards."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(batch_size, world)
    assert local_scores.shape[0] == sizes[rank], (local_scores.shape, sizes, rank)
    R = local_scores.shape[1]
    m = max(sizes)
    if local_scores.shape[0] != m:
        pad = torch.full((m - local_scores.shape[0], R), float("-inf"), dtype=local_scores.dtype, device=local_scores.device)
        local_scores = torch.cat([local_scores, pad], 0)
    out = torch.empty(world * m, R, dtype=local_scores.dtype, device=local_scores.device)
    dist.all_gather_into_tensor(out, local_scores.contiguous(), group=group)
    if all(s == m for s in sizes):
        return out
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], 0)


class PendingScores:
    """Handle of a score all-gather that has been launched but not waited for (``gather_scores_async``)."""

    def __init__(self, work, out, local, sizes, m):
        self._work, self._out, self._local, self._sizes, self._m = work, out, local, sizes, m

    def wait(self):
        """Make the current stream wait for the collective; returns the (batch_size, R) matrix."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        out, sizes, m = self._out, self._sizes, self._m
        if all(s == m for s in sizes):
            return out
        return torch.cat([out[r * m:r * m + sizes[r]] for r in range(len(sizes))], 0)


def gather_scores_async(local_scores, batch_size, group=None):
    """``gather_scores`` without waiting: the collective is ordered behind the scores on the communication stream, and the
    calling stream only waits when ``.wait()`` is called -- so the forward of the NEXT batch overlaps the all-gather of this
    one (the branch selector consumes batch i while batch i + 1 is scored).  The handle keeps the buffers alive."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(batch_size, world)
    assert local_scores.shape[0] == sizes[rank], (local_scores.shape, sizes, rank)
    R = local_scores.shape[1]
    m = max(sizes)
    if local_scores.shape[0] != m:
        pad = torch.full((m - local_scores.shape[0], R), float("-inf"), dtype=local_scores.dtype, device=local_scores.device)
        local_scores = torch.cat([local_scores, pad], 0)
    local_scores = local_scores.contiguous()
    out = torch.empty(world * m, R, dtype=local_scores.dtype, device=local_scores.device)
    work = dist.all_gather_into_tensor(out, local_scores, group=group, async_op=True)
    return PendingScores(work, out, local_scores, sizes, m)


def score_sharded(batch, score_fn, group=None):
    """Score ``batch`` (a synth.SubproblemBatch-like object with ``.slice`` and ``.batch_size``) data-parallel.

    Every rank passes the SAME full batch description and receives the (B, R) padded scores of all subproblems."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(batch.batch_size, world, rank)
    if hi > lo:
        local = score_fn(batch.slice(lo, hi))
    else:                                   # more ranks than subproblems
        R = int(batch.masks.shape[1])
        local = torch.empty(0, R, dtype=torch.float32, device=batch.masks.device)
    return gather_scores(local, batch.batch_size, group)


def decisions_from_scores(scores, relu_sizes):
    """Branch selector on the gathered matrix: first maximal score per row -> (B, 2) [layer, idx]
    (graph_score.py:41-47); rows without an undecided ReLU give [-1, -1]."""
    cum = torch.cumsum(torch.tensor(relu_sizes), 0)
    best, idx = scores.max(1)
    idx = idx.cpu()
    lay = torch.searchsorted(cum, idx, right=True)
    start = torch.cat([torch.zeros(1, dtype=cum.dtype), cum[:-1]])[lay]
    dec = torch.stack([lay, idx - start], 1)
    dec[torch.isinf(best.cpu()) & (best.cpu() < 0)] = -1
    return dec
