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


# ---- bench.py's N-rank loop (StepLoop) and launcher ------------------------------------------------------------------
class _Scores:
    def __init__(self, scores):
        self.scores = scores


def _bench_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        B, R, K = 4, 37, 5                     # per-rank shard, padded score width, steps
        calls = [0]

        def forward():                         # stand-in scorer: shard `rank` of step i holds i*1000 + global row + column/100
            i = calls[0]
            calls[0] += 1
            rows = torch.arange(rank * B, (rank + 1) * B, dtype=torch.float32).unsqueeze(1)
            s = i * 1000.0 + rows + torch.arange(R, dtype=torch.float32).unsqueeze(0) / 100.0
            s[:, ::5] = float("-inf")          # non-ambiguous columns
            return _Scores(s)
        loop = bench.StepLoop(forward, True, world * B)
        seen = []
        for i in range(K):
            loop.step()
            if loop.last_gathered is not None:
                seen.append(loop.last_gathered.clone())
        assert loop.gathers_completed == K - 1 and loop.pending is not None     # one collective still in flight
        loop.drain()
        seen.append(loop.last_gathered.clone())
        assert loop.gathers_completed == K and loop.pending is None
        # every gather holds ALL ranks' shards of ITS step, in rank order
        step_of = [int(t[0, 1].item() // 1000) for t in seen]
        for t, i in zip(seen, step_of):
            rows = torch.arange(world * B, dtype=torch.float32).unsqueeze(1)
            want = i * 1000.0 + rows + torch.arange(R, dtype=torch.float32).unsqueeze(0) / 100.0
            want[:, ::5] = float("-inf")
            assert torch.equal(t, want), (rank, i)
        assert step_of[-1] == K - 1
        # the self-description of a multi-rank run (bench.py's `dist` record): the backend's own world size, every rank's step time,
        # the bytes one all-gather moves per rank, the k_top mode
        d = bench.dist_run_description(dist, 0.010 * (rank + 1), K, forward().scores, loop.last_gathered, 1)
        assert d["process_group_world_size"] == world and d["backend"] == "gloo" and d["k_top_split"] == 1
        assert d["rank_ms_per_step"]["per_rank"] == [round(1e3 * 0.010 * (r + 1) / K, 4) for r in range(world)]
        assert d["rank_ms_per_step"]["min"] == d["rank_ms_per_step"]["per_rank"][0] and d["rank_ms_per_step"]["max"] == d["rank_ms_per_step"]["per_rank"][-1]
        assert d["gather_bytes_per_step"] == {"sent_per_rank": B * R * 4, "received_per_rank": world * B * R * 4}
        torch.save(seen[-1], os.path.join(out_dir, f"bench_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_bench_step_loop_under_gloo(tmp_path):
    """bench.py's step()/drain() with world size 2: one all-gather per step, pipelined one step deep, all K completed by
    drain(), every rank ends with the whole batch's scores."""
    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = torch.load(os.path.join(tmp_path, "bench_rank0.pt"))
    b = torch.load(os.path.join(tmp_path, "bench_rank1.pt"))
    assert torch.equal(a, b) and a.shape == (8, 37)


def test_bench_launcher_never_runs_fewer_ranks_than_asked(tmp_path):
    """`bench.py --gpus 2` without torchrun starts the ranks itself; on a box with fewer GPUs it must fail, not run 1 rank.
    Under a torchrun environment with the wrong world size it must fail too."""
    import subprocess
    import sys
    import bench
    cmd = bench.launch_command(2, ["--gpus", "2", "--steps", "3"], 29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and cmd[-5].endswith("bench.py")
    a = bench.parse_args(["--config", "4", "--gpus", "8"])
    assert (a.net, a.batch, a.gpus) == ("cifar_deep_kw", 128, 8)
    a = bench.parse_args([])
    assert (a.net, a.batch, a.gpus) == ("cifar_base_kw", 256, 1)
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box could actually run two ranks")
    root = os.path.dirname(os.path.abspath(bench.__file__))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only" in r.stderr and not r.stdout.strip()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


# ---- the branch selector on the gathered matrix (graph_score.py:41-47) ---------------------------------------
def test_branch_selector_on_the_gathered_matrix_takes_the_first_maximum():
    """`decisions_from_scores` is what every rank applies to the all-gathered (B, R) matrix.  It must give what the
    reference's `torch.max(scores, 0)` + `trans_len` walk gives per subproblem (graph_score.py:41-47, restated in the oracle's
    `decision_from_scores` on the ragged vector) -- on the reference's golden scores, and on rows with exact ties (FIRST maximal
    index: the device's k_score keeps the lowest flat index among equal scores, and N ranks must agree with one)."""
    from oracle import gnn_oracle
    from tests.common import GOLDEN_CASES, load_golden, relu_sizes
    for case in GOLDEN_CASES:
        g, batch = load_golden(case)
        sizes = relu_sizes(batch)
        for fam in ("shipped", "random"):
            padded = torch.from_numpy(g[f"{fam}_scores"])
            assert parallel.decisions_from_scores(padded, sizes).tolist() == g[f"{fam}_decisions"].tolist()
    sizes = [5, 4, 3]
    ninf = float("-inf")
    rows = torch.tensor([
        [ninf, 2.0, 2.0, ninf, 1.0, 2.0, ninf, ninf, 0.0, 2.0, ninf, ninf],       # tie inside layer 0 and across layers -> [0, 1]
        [ninf, ninf, ninf, ninf, ninf, -3.0, ninf, -3.0, ninf, ninf, -3.0, ninf],  # tie across layers 1 and 2 -> [1, 0]
        [ninf] * 12,                                                               # nothing undecided -> [-1, -1]
        [ninf] * 11 + [7.0],                                                       # the last node of the last layer -> [2, 2]
        [0.5] * 12,                                                                # all equal -> the first node
    ])
    want = [[0, 1], [1, 0], [-1, -1], [2, 2], [0, 0]]
    assert parallel.decisions_from_scores(rows, sizes).tolist() == want
    for r, w in zip(rows, want):
        mask = torch.isfinite(r).float()
        if mask.sum() > 0:
            assert gnn_oracle.decision_from_scores(r[mask != 0], mask, sizes) == w
    # sharding the rows over ranks and concatenating (what the all-gather does) cannot change a row's decision
    for world in (2, 3):
        parts = [rows[slice(*parallel.shard_bounds(len(rows), world, r))] for r in range(world)]
        assert parallel.decisions_from_scores(torch.cat(parts), sizes).tolist() == want


@pytest.mark.gpu
@pytest.mark.parametrize("net,B", [("cifar_base_kw", 64), ("cifar_deep_kw", 24)])
def test_device_decisions_equal_the_selector_on_the_scores(net, B):
    """Per-rank device decisions (k_score's argmax) == `decisions_from_scores` on the same padded scores: whether the selector
    reads the (B/G, 2) decisions or the gathered (B, R) matrix it branches on the same nodes."""
    from gnn_branching_amd import synth
    from tests.common import relu_sizes
    from tests.test_gpu_parity import make_model
    model = make_model("shipped")
    batch = synth.make_batch(net, B, seed=5)
    with torch.no_grad():
        res = model.forward_device(*batch.forward_args()).check()
    dec = parallel.decisions_from_scores(res.scores.cpu(), relu_sizes(batch))
    assert dec.tolist() == res.decisions.cpu().tolist()
    # duplicated subproblems produce bit-identical rows (batched == per-sample): the gathered matrix of a 2-rank split of
    # [batch, batch] is the concatenation, and both halves decide alike
    both = torch.cat([res.scores.cpu(), res.scores.cpu()])
    assert parallel.decisions_from_scores(both, relu_sizes(batch)).tolist() == dec.tolist() * 2
