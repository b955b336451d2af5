#!/usr/bin/env python
"""Benchmark of the GNN branching-score hot path on MI355X.

Metric (BASELINE.json): ReLU branching scores/s = subproblems x ambiguous ReLUs per second on
cifar_base_kw.  One "step" = one batched forward (T=2 rounds of message passing + score head +
per-subproblem argmax) over a batch of synthetic subproblems already resident in HBM; with N > 1
ranks each rank scores its own shard of the live subproblems and ONE RCCL all-gather collects the
padded scores for the branch selector (weak scaling: per-GPU batch fixed).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4] [--net cifar_base_kw] [--batch 256]

`--gpus N` with N > 1 and no torchrun environment starts the N ranks itself (fresh child processes of
`python -m torch.distributed.run`, spawned before this process touches the GPU) and exits with their code; under
torchrun (RANK / WORLD_SIZE set) it is one of the ranks.  A world size that does not match --gpus is an error -- it never
silently benchmarks one rank.

Prints ONE JSON line (rank 0).  The CPU oracle is imported only for the `cpu_baseline` leg.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# BASELINE.json `configs` that are bench lines (config 1 is the CPU plumbing case, config 5 needs Gurobi + CIFAR-10)
CONFIGS = {
    2: {"net": "cifar_base_kw", "batch": 256},      # the headline config: 1 x MI355X, 256 synthetic subproblems
    3: {"net": "cifar_wide_kw", "batch": 256},
    4: {"net": "cifar_deep_kw", "batch": 128},      # 1024 subproblems over 8 ranks = 128 per rank (run with --gpus 8)
}

PEAK_HBM_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3 TB/s achievable)
PEAK_F32_MFMA_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, same guide
PEAK_L2_GBS = 34500.0        # aggregate L2 -> L1 rate, same guide ("L2 (per XCD)": ~34.5 TB/s)
N_SIMD, PEAK_CLOCK_HZ = 1024, 2.4e9      # 256 CUs x 4 SIMDs; the clock the matrix peaks are quoted at
# bytes per TCP -> TCC request, calibrated on this pool (profiles/r03a_base_pmc_summary.json): k_dense_fwd_lds reads 67 MB of rows + 134 MB of
# weights = 201 MB in 1.64 M read requests -> 128 B (a cache line); k_dense_bwd_lds writes 67.1 MB in 1.05 M write requests -> 64 B
TCP_READ_REQ_BYTES, TCP_WRITE_REQ_BYTES = 128, 64
HALF_PASS_KERNELS = ("k_gather", "k_gather_update", "k_gather_input_update", "k_conv_fwd", "k_convT_bwd", "k_dense_agg", "k_prop", "k_top",
                     "k_node_update", "k_input_update")


MFMA_FLOP = 2 * 32 * 32 * 2          # one v_mfma_f32_32x32x2_f32
MLP_MACS_PER_NODE = 128 * 64 + 3 * 64 * 64   # fc3|bc3, fc3_2|bc3_1, 2nd half of fc4|bc4, fc4_2|bc4_1 (SURVEY 8(d) minus the hoisted feature chains)
AGGREGATION_KERNELS = ("k_gather", "k_gather_update", "k_conv_fwd", "k_convT_bwd", "k_dense_agg")   # edge aggregation = message passing: HBM-bound


def node_stats(batch):
    """Per ReLU layer: how many nodes are live (ub > 0), ambiguous (lb < 0 < ub) and scored (BaB mask -1) in the batch.
    The engine compacts exactly these classes on the device (k_classify); dead nodes have zero embeddings by
    definition (graph_conv.py:178/:347) and are not evaluated."""
    out, off = {}, 0
    for k in range(1, len(batch.lower_bounds_all) - 1):
        lb, ub = batch.lower_bounds_all[k].flatten(1), batch.upper_bounds_all[k].flatten(1)
        n = lb.shape[1]
        out[k] = {"nodes": int(lb.numel()), "live": int((ub > 0).sum()), "amb": int(((lb < 0) & (ub > 0)).sum()),
                  "scored": int(batch.masks[:, off:off + n].sum())}
        off += n
    return out


def plan_flops(plan, B, stats, restrict_last=True, per_launch=None):
    """Per kernel class, for ONE forward of batch B: algorithmic flops (MACs the algorithm needs for the nodes that
    are updated, sparse edge sums 2*nnz*p) and issued matrix-pipe work in fp32-MFMA equivalents x 4096 flop (one
    v_mfma_f32_32x32x2_f32 = 64 pipe cycles = 1; one v_mfma_f32_32x32x16_bf16 of the three-piece blocks = 32 cycles = 0.5;
    incl. tile padding and the zeros of the dense gather blocks): issued / time / peak = the share of the time the matrix
    pipe is busy at the 2.4 GHz the peak is quoted at."""
    T = plan["T"]
    bf3 = bool(plan.get("bf3", 0))
    W64 = 24 if bf3 else 64           # one 64x64 block: 48 bf16 MFMAs of 32 cycles, or 64 fp32 MFMAs of 64 cycles
    alg, issued, agg_bytes = {}, {}, {}
    survey = plan.setdefault("_survey_bytes", {})     # per aggregation class: SURVEY 8(d)'s letter 4 p (N_src + N_dst) per half-pass, dead rows included
    survey.clear()

    def add(d, k, v):
        d[k] = d.get(k, 0.0) + v

    def tiles(n):
        return (n + 31) // 32
    for u in plan["updates"]:
        k = u["layer"]
        nnz = u["edge_nnz"] * B
        if u["update"] == "input":
            reps, nodes = T - 1, u["nodes"] * B
            if u["kernel"] == "k_gather_input_update":
                add(alg, u["kernel"], reps * 2.0 * (2 * 64 * 64 * nodes + nnz * 64))
                add(issued, u["kernel"], reps * MFMA_FLOP * u["tiles_per_sample"] * B * (2 * u["gather_ksteps"] + W64 + 4))   # inp_b2_2 deferred, inp_b2[:, 64:] applied on the producer side
            else:
                agg, upd = u["kernel"].split("+")
                add(alg, upd, reps * 2.0 * 2 * 64 * 64 * nodes)
                add(issued, upd, reps * MFMA_FLOP * tiles(nodes) * 66)
                add(alg, agg, reps * 2.0 * nnz * 64)
            continue
        fused = "+" not in u["kernel"]              # k_gather_update: gather + node update in one launch
        for t in range(T):
            restricted = restrict_last and u["update"] == "bwd" and k == 1 and t == T - 1
            if fused and not restricted:
                agg = upd = u["kernel"]
            elif fused:                             # the restricted last step always runs as two kernels
                agg, upd = "k_gather", "k_node_update"
            else:
                agg, upd = u["kernel"].split("+")
            n_upd = stats[k]["scored"] if restricted else stats[k]["live"]
            add(alg, upd, 2.0 * MLP_MACS_PER_NODE * n_upd)
            # message passing of this half-pass: every source row read once, every updated row written once
            # (round 0 with the embedding fused into the first gather reads three scalars per source node, not a 256-B row)
            src_row_bytes = 12.0 if (plan.get("embed_fused") and u["update"] == "fwd" and k == 1 and t == 0) else 4.0 * 64
            add(agg_bytes, agg, src_row_bytes * B * u["n_src"] + 4.0 * 64 * n_upd)
            add(survey, agg, 4.0 * 64 * B * (u["n_src"] + u["nodes"]))
            if per_launch is not None:       # (bench.py's aggregate-only leg prices single launches: half-pass, its aggregation kernel class, its bytes)
                per_launch.append({"t": t, "update": u["update"], "layer": k, "agg": agg, "restricted": bool(restricted),
                                   "embed_in_gather": src_row_bytes == 12.0,
                                   "bytes": src_row_bytes * B * u["n_src"] + 4.0 * 64 * n_upd,
                                   "survey_bytes": 4.0 * 64 * B * (u["n_src"] + u["nodes"])})
            # folded chains, last layer deferred.  bf16x3 (default): a tile of nodes with r0 == r1 runs 2 blocks of 48 bf16 MFMAs (+2 small),
            # a tile that holds an ambiguous node 3; fp32 MFMA only (option bf3=0): 130 / 194 fp32 MFMAs
            live, amb = stats[k]["live"], stats[k]["amb"]
            post = u["update"] == "bwd" and k == 1 and t < T - 1 and upd in ("k_node_update", "k_gather_update")
            if upd == "k_top" and k == len(plan["sizes"]) - 2:      # layer L: one workgroup per sample, every node of the layer through the general chain
                add(issued, upd, MFMA_FLOP * B * tiles(u["nodes"]) * 194)
            elif not bf3:
                gen = tiles(n_upd) if restricted else tiles(amb)
                add(issued, upd, MFMA_FLOP * ((0 if restricted else tiles(live - amb)) * 130 + gen * 194 + (tiles(live) * 64 if post else 0)))
            elif restricted:
                add(issued, upd, MFMA_FLOP * tiles(n_upd) * (2 + 3 * W64))
            elif fused or upd == "k_top":        # (k_top: layer L-1's update on its transposed edge's live-row tiles) ambiguous nodes ride in the same tiles: share of 32-node tiles that hold at least one
                p_amb = 1.0 - (1.0 - amb / max(live, 1)) ** 32
                add(issued, upd, MFMA_FLOP * tiles(live) * (2 + (2 + p_amb) * W64 + (W64 if post else 0)))
            else:
                add(issued, upd, MFMA_FLOP * (tiles(live - amb) * (2 + 2 * W64) + tiles(amb) * (2 + 3 * W64) + (tiles(live) * W64 if post else 0)))
            frac = n_upd / max(stats[k]["nodes"], 1)
            add(alg, agg, 2.0 * nnz * 64 * frac)
            if agg in ("k_gather", "k_gather_update"):
                add(issued, agg, MFMA_FLOP * u["tiles_per_sample"] * B * 2 * u["gather_ksteps"])
            elif agg == "k_top" and u["n_src"] > 1:      # dense edge on the MFMA: row tiles x k-steps x 2 channel halves
                add(issued, agg, MFMA_FLOP * B * tiles(u["nodes"]) * ((u["n_src"] + 1) // 2) * 2)
    for k, st in stats.items():
        add(alg, "k_pre_fwd", 2.0 * (7 * 64 + 2 * 64 * 64) * st["amb"])
        add(issued, "k_pre_fwd", MFMA_FLOP * tiles(st["amb"]) * 72)
        add(alg, "k_pre_bwd", 2.0 * (7 * 64 + 3 * 64 * 64 + 192 * 64 + 64 * 64) * st["amb"])
        add(issued, "k_pre_bwd", MFMA_FLOP * tiles(st["amb"]) * 392)
        add(alg, "k_score", 2.0 * (64 * 64 + 64) * st["scored"])
        add(issued, "k_score", MFMA_FLOP * tiles(st["scored"]) * 64)
    return alg, issued, agg_bytes


def pmc_rows(pmc, cls):
    """Counter summaries of the kernel templates behind profile class `cls` (k_gather covers k_gather and k_gather16, k_gather_update
    the kernel k_gather_update_q; k_gather_scored runs under class k_gather)."""
    def same(name):
        rest = name[len(cls):] if name.startswith(cls) else None
        if cls == "k_score" and name == "k_scored_tail":      # (the tail kernel runs under profile class k_score)
            return True
        return rest is not None and (rest == "" or rest.isdigit() or rest == "_q" or (cls == "k_gather" and rest == "_scored"))
    return [v for k, v in pmc.items() if same(k) and "hbm_bytes_per_launch" in v]


def binding_bounds(kern, pmc, steps):
    """Per half-pass kernel class: which resource bounds it, from the committed counter passes of this same command
    (profiles/pmc_latest_*.json) and the live launch durations.  speed of light = max of
      hbm   HBM bytes the counters saw per launch ((2 FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction)      / 8 TB/s
      mfma  matrix-pipe busy cycles per launch (SQ_VALU_MFMA_BUSY_CYCLES, summed over the SIMDs)            / (1024 SIMDs x 2.4 GHz)
      l2    bytes the L1s exchanged with L2 per launch (TCP_TCC_READ_REQ x 128 B + TCP_TCC_WRITE_REQ x 64 B)  / 34.5 TB/s
    and `frac` = that time / the measured launch time: how close the kernel is to the bound that binds it."""
    out = {}
    if not pmc:
        return None
    for k in HALF_PASS_KERNELS:
        if k not in kern:
            continue
        rows = pmc_rows(pmc, k)
        n = sum(v.get("launches_sampled", 1) for v in rows)
        if not n:
            continue

        def mean(field):
            have = [v for v in rows if field in v]
            m = sum(v.get("launches_sampled", 1) for v in have)
            return sum(v[field] * v.get("launches_sampled", 1) for v in have) / m if m else None
        hbm, busy = mean("hbm_bytes_per_launch"), mean("SQ_VALU_MFMA_BUSY_CYCLES_per_launch")
        rd, wr = mean("TCP_TCC_READ_REQ_sum_per_launch"), mean("TCP_TCC_WRITE_REQ_sum_per_launch")
        t_us = kern[k]["avg_us"]
        terms = {"hbm": hbm / (PEAK_HBM_GBS * 1e9) * 1e6 if hbm is not None else None,
                 "mfma": busy / N_SIMD / PEAK_CLOCK_HZ * 1e6 if busy is not None else None,
                 "l2": (rd * TCP_READ_REQ_BYTES + (wr or 0.0) * TCP_WRITE_REQ_BYTES) / (PEAK_L2_GBS * 1e9) * 1e6 if rd is not None else None}
        known = {a: b for a, b in terms.items() if b is not None}
        if not known:
            continue
        bound = max(known, key=known.get)
        out[k] = {"avg_launch_us": t_us, "launches_per_step": kern[k]["launches"] // steps,
                  "sol_us": {a: round(b, 2) for a, b in known.items()}, "binding": bound,
                  "frac_of_binding_bound": round(known[bound] / t_us, 4) if t_us > 0 else None,
                  "hbm_bytes_per_launch": round(hbm) if hbm is not None else None,
                  "l1_fill_bytes_per_launch": round(rd * TCP_READ_REQ_BYTES + (wr or 0.0) * TCP_WRITE_REQ_BYTES) if rd is not None else None,
                  "mfma_busy_share": round(known["mfma"] / t_us, 4) if "mfma" in known and t_us > 0 else None}
    return out or None


def message_passing_bytes(sizes, B, T):
    """4*p*(N_src + N_dst) bytes per half-pass update src->dst, summed over the
    live half-passes (the last round's input-layer update is dead)."""
    p = 64
    L = len(sizes) - 2
    fwd = sum(sizes[k - 1] + sizes[k] for k in range(1, L + 2))
    bwd = sum(sizes[k + 1] + sizes[k] for k in range(1, L + 1))
    inp = sizes[1] + sizes[0]
    return 4.0 * p * B * (T * (fwd + bwd) + (T - 1) * inp)


class StepLoop:
    """One bench step = one forward of this rank's shard + (N > 1) the one exchange step: ONE all-gather per scored batch
    (scores -> branch selector), launched behind the scores on the communication stream.  The compute stream only waits for
    the PREVIOUS batch's gather, so the collective of batch i overlaps the forward of batch i + 1; `drain()` completes the
    last one, so every one of the K gathers finishes inside the timed region.  `forward_fn()` returns an object with a
    `.scores` (B_local, R) tensor; tests/test_parallel_gloo.py drives this class with a CPU stand-in under gloo."""

    def __init__(self, forward_fn, use_dist, total_batch):
        self.forward_fn, self.use_dist, self.total_batch = forward_fn, use_dist, total_batch
        self.pending = None
        self.last_gathered = None
        self.gathers_completed = 0
        self.status_acc = None          # set to an empty int tensor by a caller that wants the steps' status words OR-ed (bench.py --dist)

    def step(self):
        res = self.forward_fn()
        if self.status_acc is not None and hasattr(res, "status"):
            # the OR of every step's status words, kept on the device (no sync): what the forwards that ran beside a collective reported
            word = res.status[0:1]
            for c in range(1, res.status.numel()):
                word = word | res.status[c:c + 1]
            if not self.status_acc.numel():
                self.status_acc = word.clone()
            else:
                self.status_acc |= word             # (in place: one small kernel per step, no allocation)
        if self.use_dist:
            from gnn_branching_amd import parallel
            nxt = parallel.gather_scores_async(res.scores, self.total_batch)
            if self.pending is not None:
                self.last_gathered = self.pending.wait()
                self.gathers_completed += 1
            self.pending = nxt
        return res

    def drain(self):
        if self.pending is not None:
            self.last_gathered = self.pending.wait()
            self.gathers_completed += 1
            self.pending = None


def dist_run_description(dist, elapsed, steps, local_scores, gathered, top_split):
    """What a multi-GPU run says about itself in the `dist` record (all ranks must call it: one small all-gather of every rank's own wall
    time): the process group's world size as the backend reports it, each rank's ms per step (`value` uses the maximum), the bytes ONE
    score all-gather moves per rank, and which k_top mode the handles ran in.  tests/test_parallel_gloo.py runs it under gloo."""
    world = int(dist.get_world_size())
    t = torch.tensor([elapsed], dtype=torch.float64, device=local_scores.device)
    every = torch.empty(world, dtype=torch.float64, device=local_scores.device)
    dist.all_gather_into_tensor(every, t)
    rank_ms = [round(1e3 * float(x) / steps, 4) for x in every.cpu().tolist()]
    return {"process_group_world_size": world, "backend": dist.get_backend(),
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
            "gather_bytes_per_step": {"sent_per_rank": int(local_scores.numel() * local_scores.element_size()),
                                      "received_per_rank": int(gathered.numel() * gathered.element_size())},
            "k_top_split": int(top_split),
            "k_top_split_note": "1 = k_top never waits for a partner workgroup: the rule for handles that run beside a collective (DESIGN.md section 6)"}


def _status_word(res):
    v = 0
    for x in res.status.cpu().tolist():
        v |= int(x)
    return v


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n, argv, port):
    """The documented N-rank launch: one process per GPU over RCCL, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n, argv):
    """`bench.py --gpus N` outside torchrun: start the N ranks as fresh child processes (this process has not touched the
    GPU: torch.cuda.device_count() does not initialise it) and exit with their code."""
    ndev = torch.cuda.device_count()
    if ndev < n:
        print(f"bench.py: --gpus {n} requested but only {ndev} GPU(s) are visible; refusing to run fewer ranks than asked for",
              file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    env["BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(launch_command(n, argv, free_port()), env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json config: 2 = cifar_base_kw B=256 (default), 3 = cifar_wide_kw B=256, 4 = cifar_deep_kw 128 per rank "
                         "(1024 subproblems at --gpus 8)")
    ap.add_argument("--net", default=None)
    ap.add_argument("--batch", type=int, default=None, help="subproblems per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the exact-fp32 comparison leg (handle option bf3=0: exact_fp32_ms_per_step, bf3_max_abs_delta)")
    ap.add_argument("--no-aggregate-only", action="store_true", help="skip the fuse=0 leg (roofline_aggregate_only: the stand-alone edge-aggregation kernel)")
    ap.add_argument("--no-two-in-flight", action="store_true", help="skip the two-batches-in-flight leg (two_batches_in_flight: engine.BatchPipeline, throughput only)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed leg (host_fed_ms_per_step: the same batch from pinned / pageable host tensors through engine.HostFedPipeline)")
    ap.add_argument("--cpu-budget", type=float, default=75.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--dist", action="store_true",
                    help="run through torch.distributed even with --gpus 1: the process starts its rank(s) with torch.distributed.run exactly as "
                         "--gpus N > 1 does, so init_process_group('nccl'), the score all-gather and StepLoop's overlap execute at world size 1")
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config if args.config is not None else 2]
    if args.net is None:
        args.net = cfg["net"]
    if args.batch is None:
        args.batch = cfg["batch"]
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def main():
    args = parse_args()
    in_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not in_torchrun:
        if args.gpus > 1 or args.dist:
            sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
        rank, local_rank, world = 0, 0, 1
    else:
        rank = int(os.environ["RANK"])
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from gnn_branching_amd import _lib, synth
    from gnn_branching_amd.graphnet.graph_conv import GraphNet
    _lib.build_library()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the scorer)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = in_torchrun                                 # under torchrun the collective path runs even with one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ckpt = os.path.join(ROOT, "models/cifar_trained_gnn/best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    model = GraphNet(2, 64)
    model.load_state_dict(sd)
    model.eval()
    if use_dist:
        # whenever a collective can be in flight beside a forward (StepLoop overlaps the all-gather of batch i with
        # forward i + 1, and RCCL's kernel holds CUs), the handle is created with k_top's workgroup split OFF -- its split workgroups
        # spin-wait on partner workgroups that must all be resident (parallel.DIST_ENGINE_OPTIONS)
        from gnn_branching_amd import parallel
        model.engine_options = dict(parallel.DIST_ENGINE_OPTIONS)
    eng = model.engine()

    B = args.batch
    batch = synth.make_batch(args.net, B, seed=1234 + rank)
    n_amb = int(batch.masks.sum().item())

    def dev_list(ts):
        return [t.to(dev).float().contiguous() for t in ts]
    d_args = (dev_list(batch.lower_bounds_all), dev_list(batch.upper_bounds_all), dev_list(batch.dual_vars),
              dev_list(batch.primals), batch.primal_inputs.to(dev), batch.layers, batch.masks.to(dev))
    loop = StepLoop(lambda: eng.forward(*d_args), use_dist, world * B)
    if use_dist:
        loop.status_acc = torch.empty(0, dtype=torch.int32, device=dev)
    step, drain = loop.step, loop.drain

    # Untimed pre-warm (allocator, caches, the first status read), the collector's pause, a second pre-warm (clocks), then the W
    # warm-up steps the contract asks for.
    for _ in range(4):
        res = step()
    torch.cuda.synchronize()
    res.check()                 # (its device->host copy sits before the warm-up steps: the first launch after one has been seen to stall)
    sizes = eng.sizes
    # The host only enqueues (0.2 ms per step against ~1 ms of GPU work), but a full collection of Python's cyclic GC walks
    # every object torch and numpy created at import -- 35-100 ms, i.e. several times the whole K-step region -- and fires at
    # an allocation count, i.e. at a fixed step index (tools/host_stall2.py: step 52 for base, 7 and 99 for deep).  Standard
    # benchmarking hygiene: collect now, keep the collector off while the K steps are timed.  The collection comes BEFORE the W
    # warm-up steps: behind them its 35-100 ms of idle GPU let the clocks drop, and the first timed steps paid for ramping them
    # up again (base: 0.789 ms per step at K = 100 against 0.774 at K = 1000 on one box; the default K is 20).
    import gc
    late = bool(os.environ.get("BENCH_GC_LATE"))       # dev A/B: the earlier order (collect between the warm-up steps and the timed region)
    if not late:
        gc.collect()
        gc.disable()
    prewarm = int(os.environ.get("BENCH_PREWARM", "64"))
    for _ in range(prewarm):
        res = step()
    for _ in range(args.warmup):
        res = step()
    if late:
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()

    def sync():
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    trace = [] if os.environ.get("BENCH_TRACE") else None      # dev: host time of every timed step() call
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if trace is not None:
            ta = time.perf_counter()
        res = step()
        if trace is not None:
            trace.append(round(1e3 * (time.perf_counter() - ta), 2))
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if trace is not None:
        print("host ms per timed step():", trace, file=sys.stderr)
    run_desc = None
    if use_dist:
        run_desc = dist_run_description(dist, elapsed, args.steps, res.scores, loop.last_gathered, eng.get_option("top_split"))
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_amb], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        total_amb = float(tot.item())
    else:
        total_amb = float(n_amb)
    res.check()
    dist_record = None
    if use_dist:
        # the collective path against the plain one: the rows this rank received for its own shard in the LAST all-gather (launched while the
        # next forward was being enqueued) must be the bits a forward with no collective in flight produces, and no kernel may have raised a
        # status bit (k_top's split workgroups wait for each other while RCCL's kernel holds CUs)
        plain = eng.forward(*d_args)
        torch.cuda.synchronize()
        plain.check()
        got = loop.last_gathered[rank * B:(rank + 1) * B]
        same = bool(torch.equal(got, plain.scores))
        assert same, "all-gathered scores differ from the non-distributed forward"
        assert loop.gathers_completed >= args.steps, (loop.gathers_completed, args.steps)
        dist_record = {"world_size": world, **run_desc,      # (what the first real multi-GPU run needs to be self-describing)
                       "gathers_completed": loop.gathers_completed,
                       "gathered_rows": int(loop.last_gathered.shape[0]), "gathered_equals_plain_forward_bitwise": same,
                       # the OR over every step of the run -- the forwards that overlapped an all-gather -- and the plain forward's
                       "status_word": int(_status_word(plain)) | (int(loop.status_acc.cpu()[0]) if loop.status_acc is not None and loop.status_acc.numel() else 0),
                       "self_launched": bool(os.environ.get("BENCH_SELF_LAUNCHED"))}

        if not args.no_two_in_flight:
            try:
                dist_record["dist_two_in_flight"] = dist_two_in_flight_leg(sd, d_args, plain, max(args.steps, 100), world * B, rank, B,
                                                                           round(1e3 * elapsed / args.steps, 4))
            except Exception as e:      # noqa: BLE001
                dist_record["dist_two_in_flight"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- per-kernel durations with HIP events on the launch stream (same K steps, instrumented) ----
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward(*d_args)
    torch.cuda.synchronize()
    instrumented = time.perf_counter() - t1
    prof = eng.profile_read(reset=True)
    eng.profile_enable(False)

    if rank == 0:
        T = model.T
        kern = {k: {"ms_total": round(v[0], 4), "launches": int(v[1]), "avg_us": round(1e3 * v[0] / max(v[1], 1), 2)}
                for k, v in prof.items() if v[1]}
        dom = max(kern, key=lambda k: kern[k]["ms_total"])
        plan = eng.describe()
        stats = node_stats(batch)
        alg, issued, agg_bytes = plan_flops(plan, B, stats)
        dom_s = prof[dom][0] * 1e-3 / args.steps            # seconds of the dominant kernel class per forward
        # HBM traffic of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
        # (tools/pmc.sh + tools/pmc_table.py: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, gfx950 correction applied)
        traffic = None
        pmc = None
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_latest_{args.net}_B{B}.json")
        if not os.path.exists(pmc_path) and args.net == "cifar_base_kw" and B == 256:
            pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            # a profile class may cover several kernel templates (k_gather, k_gather16): launch-weighted mean
            pmc = json.load(open(pmc_path))
            rows = pmc_rows(pmc, dom)
            n = sum(v.get("launches_sampled", 1) for v in rows)
            traffic = round(sum(v["hbm_bytes_per_launch"] * v.get("launches_sampled", 1) for v in rows) / n) if n else None
        ach_tf = alg.get(dom, 0.0) / dom_s / 1e12 if dom_s > 0 else 0.0
        iss_tf = issued.get(dom, 0.0) / dom_s / 1e12 if dom_s > 0 else 0.0
        launches = kern[dom]["launches"] // args.steps
        if dom in AGGREGATION_KERNELS:
            # an edge-aggregation (message-passing) kernel: HBM-bound by SURVEY 8(d).  Algorithmic bytes per launch =
            # 4*p*(source rows read once + updated destination rows written once), averaged over the class's launches.
            ach_gbs = agg_bytes.get(dom, 0.0) / dom_s / 1e9 if dom_s > 0 else 0.0
            sv_gbs = plan.get("_survey_bytes", {}).get(dom, 0.0) / dom_s / 1e9 if dom_s > 0 else 0.0
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach_gbs / PEAK_HBM_GBS, 4), "traffic": traffic,
                        # frac: bytes the launch must move (live source rows + updated rows; round 0 reads 12 B per source node);
                        # frac_survey_bytes: SURVEY 8(d)'s letter, 4 p (N_src + N_dst) per half-pass with dead rows counted, over the same time
                        "frac_survey_bytes": round(sv_gbs / PEAK_HBM_GBS, 4),
                        "avg_launch_us": kern[dom]["avg_us"], "launches_per_step": launches,
                        "algorithmic_bytes_per_launch": round(agg_bytes.get(dom, 0.0) / max(launches, 1)),
                        "issued_mfma_tflops": round(iss_tf, 2), "issued_mfma_frac": round(iss_tf / PEAK_F32_MFMA_TFLOPS, 4)}
        else:
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach_tf, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach_tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                        "avg_launch_us": kern[dom]["avg_us"], "launches_per_step": launches,
                        "algorithmic_gflop_per_step": round(alg.get(dom, 0.0) / 1e9, 2),
                        "issued_mfma_tflops": round(iss_tf, 2), "issued_mfma_frac": round(iss_tf / PEAK_F32_MFMA_TFLOPS, 4)}
        # the node-update class next to it (fp32 MFMA bound), whichever of the two is dominant
        nu_s = prof.get("k_node_update", (0.0, 0))[0] * 1e-3 / args.steps
        roofline_nu = None
        if nu_s > 0:
            nu_tf = alg.get("k_node_update", 0.0) / nu_s / 1e12
            nu_iss = issued.get("k_node_update", 0.0) / nu_s / 1e12
            # frac = matrix-pipe time actually issued (after the folds; a 32-cycle bf16 MFMA of a three-piece block counts as half
            # an fp32 MFMA) over the kernel time: <= 1 by construction.  The reference's MACs for the same nodes / time is kept
            # beside it as `reference_equivalent_tflops`: it exceeds the fp32 peak because the folds remove 40-60 % of those MACs.
            roofline_nu = {"kernel": "k_node_update", "bound": "mfma", "achieved": round(nu_iss, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s (issued, fp32-MFMA equivalents)", "frac": round(nu_iss / PEAK_F32_MFMA_TFLOPS, 4),
                           "reference_equivalent_tflops": round(nu_tf, 2)}
        # message passing is fused into the update kernels (the aggregate never reaches HBM): its algorithmic bytes
        # 4*p*(N_src+N_dst) per half-pass over the time of every kernel that performs an update
        mp_names = HALF_PASS_KERNELS
        # the restricted last half-pass (scored gather + update of layer 1) runs inside k_scored_tail (profile class k_score) on the
        # default path: its whole time is counted (score head included: conservative), as the three kernels' gather + update were
        if "k_gather" not in kern and "k_node_update" not in kern and "k_score" in kern:
            mp_names = HALF_PASS_KERNELS + ("k_score",)
        mp_ms = sum(prof[k][0] for k in mp_names if k in prof)
        mp_bytes = message_passing_bytes(sizes, B, T) * args.steps
        mp_gbs = mp_bytes / (mp_ms * 1e-3) / 1e9 if mp_ms > 0 else 0.0
        # counter-based: HBM bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 correction) of the same kernels per forward
        mp_traffic = None
        if pmc is not None:
            tot, ok = 0.0, True
            for k in mp_names:
                if k in kern:
                    rows = pmc_rows(pmc, k)
                    n = sum(v.get("launches_sampled", 1) for v in rows)
                    if not n:
                        ok = False
                        break
                    tot += sum(v["hbm_bytes_per_launch"] * v.get("launches_sampled", 1) for v in rows) / n * (kern[k]["launches"] // args.steps)
            mp_traffic = round(tot) if ok else None
        mp_s = mp_ms * 1e-3 / args.steps
        roofline_mp = {"kernels": "all half-pass kernels (edge aggregation + node update)" + (" incl. k_scored_tail" if "k_score" in mp_names else ""), "bound": "hbm",
                       "achieved": round(mp_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(mp_gbs / PEAK_HBM_GBS, 4),
                       "traffic": mp_traffic,
                       "frac_counter": round(mp_traffic / mp_s / 1e9 / PEAK_HBM_GBS, 4) if (mp_traffic and mp_s > 0) else None,
                       "us_per_step": round(1e3 * mp_ms / args.steps, 1),
                       "bytes_per_subproblem": message_passing_bytes(sizes, 1, T),
                       "note": "achieved = SURVEY 8(d) algorithmic bytes 4*p*(N_src + N_dst) per half-pass / time of all half-pass kernels; "
                               "frac_counter = HBM bytes the counters saw (dead rows are skipped, round 0 computes its source rows) / the same time"}
        # ---- the same batch with every block on the exact-fp32 MFMA (handle option bf3=0 on a second handle),
        # outside the timed region of the headline: what the three-piece bf16 blocks buy, and how far their scores are from it.
        # A side leg never takes the headline down with it: a failure is recorded in the JSON (`side_leg_errors`).
        side_errors = {}
        # ---- the same batch arriving as HOST tensors (the reference pays its H2D copies inside the call, graph_score.py:26-30): the copies of
        # batch i + 1 under the forward of batch i (engine.HostFedPipeline).  Never `value`: reported beside it.
        # First of the side legs: a pipeline built right after other handles and their workspaces were released runs its first ~100 steps at 1.3-2.3 ms
        # (tools/hostfed_repeat.py: the first pipeline after gc.collect() + empty_cache() 1.69 ms, the next one 0.92 again) -- allocation churn, not the copies.
        host_fed = None
        if not args.no_host_fed:
            try:
                host_fed = host_fed_leg(eng, batch, max(args.steps, 100), round(1e3 * elapsed / args.steps, 4), res)
            except Exception as e:      # noqa: BLE001
                side_errors["host_fed"] = f"{type(e).__name__}: {e}"
        two_in_flight = None
        if not args.no_two_in_flight and not use_dist:
            try:
                two_in_flight = two_in_flight_leg(sd, d_args, res, max(args.steps, 200), 1e3 * elapsed / args.steps, total_amb)
            except Exception as e:      # noqa: BLE001
                side_errors["two_in_flight"] = f"{type(e).__name__}: {e}"
        exact_ms = bf3_delta = None
        if plan.get("bf3") and not args.no_exact_fp32:
            try:
                exact_ms, bf3_delta = exact_fp32_leg(sd, d_args, res, max(args.steps, 100))
            except Exception as e:      # noqa: BLE001
                side_errors["exact_fp32"] = f"{type(e).__name__}: {e}"
        # ---- the edge aggregation ALONE ("standalone message-passing (aggregate-only) kernel for the HBM-roofline
        # measurement, plus the fused production variant"): the same batch through a second handle with option fuse=0, where every conv
        # half-pass is k_gather (aggregate rows -> HBM) + k_node_update; identical scores.  Outside the timed region of the headline.
        agg_only = None
        if not args.no_aggregate_only:
            try:
                agg_only = aggregate_only_leg(sd, d_args, res, args, batch, stats, B)
            except Exception as e:      # noqa: BLE001
                side_errors["aggregate_only"] = f"{type(e).__name__}: {e}"
                agg_only = {"error": side_errors["aggregate_only"]}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(sd, args.net, args.cpu_budget)
        out = {
            "metric": "ReLU branching scores/sec (subproblems x ambiguous-ReLUs/s)",
            "value": round(total_amb * args.steps / elapsed, 1),
            "unit": "scores/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # what ran, untimed, before the K timed steps besides the W warm-up steps the contract asks for: 4 steps (allocator, first status
            # read) + BENCH_PREWARM steps (clocks); Python's cyclic collector is run once before them and kept off while the K steps are timed
            "prewarm_steps": 4 + prewarm, "gc_disabled": True,
            # handle options the timed engine was created with (include/gnnb.h gnnb_set_option; {} = the library's defaults)
            "engine_options": dict(eng.options),
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (bf16x3-split 64x64 blocks)" if plan.get("bf3") else "f32", "data": "synthetic",
            "config": {"workload": f"{args.net}, {B} synthetic subproblems per rank x {world} rank(s) = {world * B}, T=2, p=64, shipped cifar_trained_gnn weights",
                       "batch_per_rank": B, "ranks": world, "global_batch": world * B,
                       "subproblems_per_s": round(world * B * args.steps / elapsed, 1),
                       "ambiguous_per_subproblem": round(total_amb / (world * B), 1),
                       "parallelism": f"dp{world}" + (" + 1 all-gather(scores)/step" if world > 1 else "")},
            "roofline": roofline,
            "roofline_node_update": roofline_nu,
            "roofline_message_passing": roofline_mp,
            "roofline_aggregate_only": agg_only,
            "binding_bounds": binding_bounds(kern, pmc, args.steps),
            "exact_fp32_ms_per_step": round(exact_ms, 4) if exact_ms is not None else None,
            "bf3_max_abs_delta": bf3_delta,
            "cpu_baseline": cpu,
            "kernels": kern,
            "plan": plan["updates"],
            "node_classes": {str(k): v for k, v in stats.items()},
            "instrumented_ms_per_step": round(1e3 * instrumented / args.steps, 4),
            "host_fed_ms_per_step": host_fed,
            "two_batches_in_flight": two_in_flight,
            "dist": dist_record,
            "side_leg_errors": side_errors or None,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def exact_fp32_leg(sd, d_args, res, steps):
    from gnn_branching_amd.graphnet.graph_conv import GraphNet
    model32 = GraphNet(2, 64)
    model32.load_state_dict(sd)
    model32.engine_options = {"bf3": 0}           # a second handle with every block on the exact-fp32 MFMA
    eng32 = model32.eval().engine()
    for _ in range(40):           # (a new handle: bind, allocations -- the GPU idled meanwhile and its clocks dropped)
        r32 = eng32.forward(*d_args)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(steps):
        r32 = eng32.forward(*d_args)
    torch.cuda.synchronize()
    exact_ms = 1e3 * (time.perf_counter() - t2) / steps
    r32.check()
    a32, a16 = r32.scores.cpu().numpy(), res.scores.cpu().numpy()
    fin = np.isfinite(a32)
    if not np.array_equal(fin, np.isfinite(a16)):
        raise RuntimeError("exact-fp32 leg: different set of scored nodes")
    return exact_ms, (float(np.abs(a32[fin] - a16[fin]).max()) if fin.any() else 0.0)


def host_fed_leg(eng, batch, steps, device_resident_ms, res):
    """ms per step with the batch arriving as host tensors every step, copies overlapped with the previous step's forward
    (engine.HostFedPipeline: two device buffer sets, a copy stream): once from pinned host tensors, once from pageable ones."""
    from gnn_branching_amd.engine import HostFedPipeline
    args = list(batch.forward_args())
    nbytes = 4 * sum(t.numel() for g in args if isinstance(g, list) for t in g) + 4 * (args[4].numel() + args[6].numel())
    out = {"device_resident": device_resident_ms, "bytes_per_batch": int(nbytes)}
    # pinned / pageable: the default pipeline (dual_vars and primals cross the link as records of the ambiguous nodes only);
    # pinned_whole_tensors: every tensor whole (round 4's pipeline, HostFedPipeline(compact=False))
    for kind in ("pinned", "pageable", "pinned_whole_tensors"):
        a = [[t.float().contiguous() for t in g] if isinstance(g, list) else g for g in args]
        a[4], a[6] = args[4].float().contiguous(), args[6].float().contiguous()
        if kind.startswith("pinned"):
            a = [[t.pin_memory() for t in g] if isinstance(g, list) else g for g in a]
            a[4], a[6] = a[4].pin_memory(), a[6].pin_memory()
        pipe = HostFedPipeline(eng, compact=(kind != "pinned_whole_tensors"))
        for _ in range(24):
            r = pipe.submit(*a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = pipe.submit(*a)
        torch.cuda.synchronize()
        out[kind] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
        if kind == "pinned":
            out["link_bytes_per_batch"] = int(pipe.link_bytes)
        r.check()
        if not torch.equal(r.scores, res.scores):
            raise RuntimeError(f"host-fed ({kind}) scores differ from the device-resident forward")
    out["pinned_over_device_resident"] = round(out["pinned"] / device_resident_ms, 3)
    # what the link gives on this box (pinned 2 MB pieces back to back, nothing else running) against what hiding the copies needs
    src = torch.empty(1 << 19, dtype=torch.float32).pin_memory()
    dst = torch.empty(32, 1 << 19, dtype=torch.float32, device=res.scores.device)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(32):
            dst[j].copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        link = 32 * (1 << 21) / (time.perf_counter() - t0) / 1e9
    out["h2d_link_GBps"] = round(link, 1)
    out["h2d_GBps_needed_to_hide"] = round(out["link_bytes_per_batch"] / (device_resident_ms * 1e-3) / 1e9, 1)
    out["h2d_GBps_pinned_achieved"] = round(out["link_bytes_per_batch"] / (out["pinned"] * 1e-3) / 1e9, 1)
    out["note"] = ("one step = submit(batch from host tensors): H2D of batch i+1 (2 MB pieces; dual_vars / primals as records of the ambiguous nodes, "
                   "link_bytes_per_batch of the bytes_per_batch the tensors hold) on a copy stream under the forward of batch i; pageable "
                   "inputs go through the runtime's staging.  The copies hide completely only where the link sustains h2d_GBps_needed_to_hide "
                   "(bytes_per_batch / device-resident step); below it the step is the copy time")
    return out


def two_in_flight_leg(sd, d_args, res, steps, one_ms, n_amb):
    """Throughput with TWO independent batches in flight (engine.BatchPipeline: two handles, two streams, the same batch shape dealt to
    them in turn) -- never `value`: SURVEY 8(d) defines the metric on the wall time of ONE batched forward."""
    from gnn_branching_amd.engine import BatchPipeline
    pipe = BatchPipeline(sd, depth=2)
    import gc
    gc.collect()
    gc.disable()                  # (as for the headline's timed region: a full collection costs tens of ms -- and BEFORE the warm-up, see there)
    for _ in range(48):
        r = pipe.submit(*d_args)
    pipe.synchronize()
    torch.cuda.synchronize()
    try:
        t0 = time.perf_counter()
        rs = []
        for _ in range(steps):
            rs = (rs + [pipe.submit(*d_args)])[-2:]       # (only the last result of each slot is kept: holding all of them makes every submit a fresh device allocation)
        pipe.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
    finally:
        gc.enable()
    for r in rs[-2:]:
        r.check()
        if not torch.equal(r.scores, res.scores):
            raise RuntimeError("scores of the two-in-flight pipeline differ from the plain forward")
    return {"ms_per_batch": round(ms, 4), "scores_per_s": round(n_amb / (ms * 1e-3), 1), "over_one_in_flight": round(ms / one_ms, 3),
            "note": "two independent batches in flight on two streams (two handles: own workspaces; k_top's workgroup split off): the ramps, tails "
                    "and launch gaps of one batch's dependent launches fill with the other's kernels.  Throughput only -- a batch takes longer "
                    "from submit to ready; `value` stays one batched forward at a time"}


def dist_two_in_flight_leg(sd, d_args, plain, steps, total_batch, rank, B, one_handle_ms):
    """The multi-GPU path with TWO handles in flight (engine.BatchPipeline: two handles, two streams) and the score all-gather of every
    batch launched behind its forward on that batch's OWN stream -- so forward i + 1 (other handle, other stream) and the collective of
    batch i overlap, and the caller's stream stays idle.  Never `value` (the metric is one batched forward at a time);
    reported in `dist` beside the one-handle figure.  Asserts the gathered rows of this rank's shard == the plain forward bitwise and
    a clean status word over every forward of the leg."""
    import gc
    from gnn_branching_amd import parallel
    from gnn_branching_amd.engine import BatchPipeline
    pipe = BatchPipeline(sd, depth=2)
    pend = [None, None]
    state = {"done": 0, "last": None, "status": [None] * pipe.depth}      # one status accumulator per slot: each is only touched on its slot's stream

    def step():
        k = pipe.i % pipe.depth
        st = pipe.streams[k]
        if pend[k] is not None:
            with torch.cuda.stream(st):
                state["last"] = pend[k].wait()          # the slot's stream waits for its previous collective (bounds what is in flight)
            state["done"] += 1
        res = pipe.submit(*d_args)
        with torch.cuda.stream(st):
            pend[k] = parallel.gather_scores_async(res.scores, total_batch)      # ordered behind THIS forward on its stream
            word = res.status[0:1]
            for c in range(1, res.status.numel()):
                word = word | res.status[c:c + 1]
            state["status"][k] = word.clone() if state["status"][k] is None else (state["status"][k] | word)
        return res

    def drain():
        for k in range(pipe.depth):
            if pend[k] is not None:
                with torch.cuda.stream(pipe.streams[k]):
                    state["last"] = pend[k].wait()
                state["done"] += 1
                pend[k] = None
        pipe.synchronize()
        torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        for _ in range(48):
            step()
        drain()
        state["status"] = [None] * pipe.depth
        n0 = state["done"]
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
        ms = 1e3 * (time.perf_counter() - t0) / steps
    finally:
        gc.enable()
    got = state["last"][rank * B:(rank + 1) * B]
    same = bool(torch.equal(got, plain.scores))
    status = 0
    for acc in state["status"]:        # (after drain(): every stream is synchronised, the accumulators are final)
        if acc is not None:
            status |= int(acc.cpu()[0])
    if not same or status != 0:
        raise RuntimeError(f"two handles in flight + all-gather: gathered == plain forward {same}, status word {status}")
    return {"ms_per_batch": round(ms, 4), "one_handle_ms_per_step": one_handle_ms, "over_one_handle": round(ms / one_handle_ms, 3),
            "gathers_completed": state["done"] - n0, "gathered_equals_plain_forward_bitwise": same, "status_word": status,
            "note": "engine.BatchPipeline (two handles, two streams, k_top's workgroup split off) with each batch's all-gather launched behind its "
                    "forward on that batch's own stream; throughput only, never `value`"}


def restricted_source_rows(batch, k):
    """Rows the restricted last step's aggregate of ReLU layer k has to read: the LIVE nodes of layer k + 1 that lie in the
    transposed-conv window of at least one scored node of layer k (graph_conv.py:299-318 evaluated where the score head looks)."""
    import torch.nn.functional as F
    affine = [l for l in batch.layers["fixed_layers"] if isinstance(l, (torch.nn.Conv2d, torch.nn.Linear))]
    conv = affine[k]                               # edge k + 1: layer k -> layer k + 1
    if not isinstance(conv, torch.nn.Conv2d):
        return None
    shape = batch.lower_bounds_all[k].shape        # (B, C, H, W)
    off = sum(int(np.prod(t.shape[1:])) for t in batch.lower_bounds_all[1:k])
    n = int(np.prod(shape[1:]))
    scored = batch.masks[:, off:off + n].reshape(shape).amax(1, keepdim=True).float()
    kh, kw = conv.kernel_size
    need = F.conv2d(scored, torch.ones(1, 1, kh, kw), stride=conv.stride, padding=conv.padding) > 0
    live = batch.upper_bounds_all[k + 1] > 0
    return int((live & need).sum())


def aggregate_only_leg(sd, d_args, res, args, batch, stats, B):
    """roofline_aggregate_only: the stand-alone edge-aggregation launches of one forward, priced three ways each --
      frac_rows_moved    256 B x (LIVE source rows the launch has to read + destination rows it writes) / time / 8 TB/s
      frac_survey_bytes  SURVEY 8(d)'s 4 p (N_src + N_dst) (dead rows counted too)                      / time / 8 TB/s
      frac_counter       HBM bytes the counters saw for that kernel template ((2 FETCH + WRITE) KB)     / time / 8 TB/s  (committed
                         PMC passes of this leg: profiles/pmc_latest_<net>_B<B>_aggonly.json; null without them)."""
    from gnn_branching_amd.graphnet.graph_conv import GraphNet
    modelA = GraphNet(2, 64)
    modelA.load_state_dict(sd)
    # every conv half-pass as k_gather (aggregate rows -> HBM) + k_node_update, round 0's rows from k_embed, and the restricted last
    # step's aggregate as its own launch too
    modelA.engine_options = {"fuse": 0, "embed_fuse": 0, "tail_max_b": 0}
    engA = modelA.eval().engine()
    for _ in range(40):           # (a new handle: the GPU idled while it was built and its clocks dropped)
        rA = engA.forward(*d_args)
    torch.cuda.synchronize()
    engA.profile_enable(True)
    engA.profile_read(reset=True)
    engA.profile_trace(65536)
    for _ in range(args.steps):
        rA = engA.forward(*d_args)
    torch.cuda.synchronize()
    profA = engA.profile_read(reset=True)
    traceA = [ms for name, ms in engA.profile_trace(65536) if name == "k_gather"]
    engA.profile_enable(False)
    rA.check()
    finA = torch.isfinite(res.scores)
    same_set = bool(torch.equal(finA, torch.isfinite(rA.scores)))
    agg_delta = float((rA.scores[finA] - res.scores[finA]).abs().max()) if (same_set and finA.any()) else None
    same_dec = bool(torch.equal(rA.decisions, res.decisions))
    # parity of this leg against the default path is asserted here (into side_leg_errors / the leg's "error" field, never into `value`):
    # the same scored set, the same decisions, scores within the 2e-5 two fp32-grade evaluations of the same sums differ by
    if not same_set or not same_dec or agg_delta is None or agg_delta > 2e-5:
        raise RuntimeError(f"aggregate-only leg disagrees with the default path: same scored set {same_set}, same decisions {same_dec}, max |delta| {agg_delta}")
    planA = engA.describe()
    launchesA = []
    plan_flops(planA, B, stats, per_launch=launchesA)
    by_key = {(u["update"], u["layer"]): u for u in planA["updates"]}
    # the stand-alone aggregates of one forward in launch order (forward sweep up, backward sweep down, per round)
    launchesA = sorted((x for x in launchesA if x["agg"] == "k_gather"),
                       key=lambda x: (x["t"], 0 if x["update"] == "fwd" else 1, x["layer"] if x["update"] == "fwd" else -x["layer"]))
    pmcA = None
    pmc_path = os.path.join(ROOT, "profiles", f"pmc_latest_{args.net}_B{B}_aggonly.json")
    if os.path.exists(pmc_path):
        pmcA = json.load(open(pmc_path)).get("_templates")
    per, tot_moved, tot_counter, counter_ok = [], 0.0, 0.0, pmcA is not None
    if launchesA and len(traceA) == len(launchesA) * args.steps:
        for i, x in enumerate(launchesA):
            us = 1e3 * sum(traceA[i::len(launchesA)]) / args.steps
            u = by_key[(x["update"], x["layer"])]
            src = x["layer"] - 1 if x["update"] == "fwd" else x["layer"] + 1
            n_upd = stats[x["layer"]]["scored"] if x["restricted"] else stats[x["layer"]]["live"]
            if src == 0:
                src_rows, what = B * u["n_src"], "dense source (input layer: every row live)"
            elif x["restricted"]:
                r = restricted_source_rows(batch, x["layer"])
                src_rows = r if r is not None else stats[src]["live"]
                what = "restricted last step (scored nodes only; live rows inside their windows)"
            else:
                src_rows, what = stats[src]["live"], "live rows of a ReLU layer (sparse walk)"
            moved = 256.0 * (src_rows + n_upd)
            lanes = u.get("tile_nodes", 32)
            tmpl = ("k_gather16" if lanes == 16 else "k_gather") + ("<false, false>" if src == 0 else "<false, true>")
            counter = None
            if pmcA is not None:
                row = (pmcA.get("k_gather_scored") if x["restricted"] and "k_gather_scored" in pmcA else None) or pmcA.get(tmpl)
                counter = row.get("hbm_bytes_per_launch") if row else None
                if row and x["restricted"] and "k_gather_scored" in pmcA:
                    tmpl = "k_gather_scored"
            counter_ok = counter_ok and counter is not None
            tot_moved += moved
            tot_counter += counter or 0.0
            per.append({"half_pass": f"{x['update']} layer {x['layer']} round {x['t']}", "kernel": tmpl, "source": what, "avg_us": round(us, 2),
                        "source_rows_read": int(src_rows), "rows_written": int(n_upd),
                        "frac_rows_moved": round(moved / us / 1e3 / PEAK_HBM_GBS, 4),
                        "frac_survey_bytes": round(x["survey_bytes"] / us / 1e3 / PEAK_HBM_GBS, 4),
                        "frac_counter": round(counter / us / 1e3 / PEAK_HBM_GBS, 4) if counter else None,
                        "counter_over_rows_moved": round(counter / moved, 3) if counter else None})
    msA, nA = profA.get("k_gather", (0.0, 0))
    if not nA:
        return {"error": "no k_gather launch in the fuse=0 run"}
    strict = sum(planA["T"] * 4.0 * 64 * B * (u["n_src"] + u["nodes"]) for u in planA["updates"]
                 if u["update"] != "input" and u["kernel"].split("+")[0] == "k_gather")
    sA = msA * 1e-3 / args.steps
    return {"kernel": "k_gather (edge aggregate alone: handle options fuse=0 embed_fuse=0 tail_max_b=0 -- rows to HBM, node update, input embedding and score head in their own launches)",
            "bound": "hbm", "same_scored_set_as_default_path": same_set, "same_decisions_as_default_path": same_dec,
            "max_abs_score_delta_vs_default_path": agg_delta,
            "avg_launch_us": round(1e3 * msA / nA, 2), "launches_per_step": int(nA // args.steps), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac_rows_moved": round(tot_moved / sA / 1e9 / PEAK_HBM_GBS, 4) if per else None,
            "frac_survey_bytes": round(strict / sA / 1e9 / PEAK_HBM_GBS, 4),
            "frac_counter": round(tot_counter / sA / 1e9 / PEAK_HBM_GBS, 4) if (per and counter_ok) else None,
            "achieved_rows_moved": round(tot_moved / sA / 1e9, 1) if per else None, "achieved_survey_bytes": round(strict / sA / 1e9, 1),
            "per_launch": per, "best_launch_frac_rows_moved": max((x["frac_rows_moved"] for x in per), default=None),
            "note": "frac_rows_moved: 256 B x (live source rows the launch must read + rows it writes) / time; frac_survey_bytes: SURVEY 8(d)'s "
                    "4 p (N_src + N_dst) per half-pass (dead rows counted too) / the same time; frac_counter: HBM bytes of the committed PMC passes of "
                    "this leg ((2 FETCH_SIZE + WRITE_SIZE) KB per launch of the kernel template) / the same time"}


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, net, budget_s=75.0):
    """The CPU oracle (oracle/gnn_oracle.py: a torch-CPU port with the reference's aten op sequence, pinned by the reference's
    own outputs) timed on this box's host cores, by the protocol of SURVEY 8(d): batch sizes 1 / 16 / 256 (or the largest
    that fits the time budget), all host threads, 16 threads and ONE thread (the reference's own deployment pins one core with taskset,
    scripts/bab_mip.sh:3-5,40), each with torch's default denormal handling (flush-to-zero off: the shipped
    checkpoint holds 21 all-subnormal tensors, which put x86 cores on their slow path) and with torch.set_flush_denormal(True)
    (scores are identical: those tensors contribute nothing at fp32).  Every cell: 2 warm-up forwards, then the median of 5
    timed ones where the budget allows (cells that would not fit run fewer repetitions and say so).  `value` is the best cell.
    A bounded sample: about `budget_s` seconds of CPU work in all."""
    from gnn_branching_amd import synth
    from oracle import gnn_oracle
    state = {k: v.numpy() for k, v in sd.items()}
    all_threads = torch.get_num_threads()
    ftz_default = False            # torch never enables flush-to-zero by itself (this torch has no getter; set_flush_denormal is the only switch)
    batches = {}

    def get_batch(b):
        if b not in batches:
            big = synth.make_batch(net, b, seed=1234)
            batches[b] = (big, int(big.masks.sum().item()))
        return batches[b]

    def one(b):
        t0 = time.perf_counter()
        with torch.no_grad():
            gnn_oracle.oracle_forward(state, *b.forward_args())
        return time.perf_counter() - t0

    cells, skipped = [], []
    # all host threads, 16 (the small per-layer ops of this path stop scaling there: on a 128-thread host all threads are
    # 5-10x SLOWER than 16), and one
    thread_sets = sorted({all_threads, min(16, all_threads), 1}, reverse=True)
    n_cells = 2 * len(thread_sets) * 3
    per_cell = budget_s / n_cells
    t_start = time.perf_counter()
    try:
        for ftz in (False, True):
            if not torch.set_flush_denormal(ftz) and ftz:
                continue                                   # this CPU has no flush-to-zero mode
            for nt in thread_sets:
                torch.set_num_threads(nt)
                per_sub = None                             # seconds per subproblem, from the previous (smaller) batch size
                for want_b in (1, 16, 256):
                    b = want_b
                    while per_sub is not None and b > 16 and per_sub * b * 3 > per_cell:
                        b //= 2                            # "256 or the largest that fits"
                    if b != want_b and b <= 16:
                        skipped.append({"batch": want_b, "threads": nt, "flush_denormal": ftz,
                                        "why": f"a forward of {want_b} would take ~{per_sub * want_b:.1f}s of a {per_cell:.1f}s cell"})
                        continue
                    batch, n_amb = get_batch(b)
                    t_first = one(batch)                   # warm-up 1
                    runs, warm = [], 1
                    if t_first * 3 <= per_cell:
                        one(batch)                         # warm-up 2
                        warm = 2
                        reps = int(max(1, min(5, (per_cell - 2 * t_first) // max(t_first, 1e-6))))
                        runs = [one(batch) for _ in range(reps)]
                    else:
                        runs = [t_first]                   # over budget: the single (cold) forward is the sample
                        warm = 0
                    med = statistics.median(runs)
                    per_sub = med / b
                    cells.append({"batch": b, "threads": nt, "flush_denormal": ftz, "ms_per_forward": round(1e3 * med, 2),
                                  "scores_per_s": round(n_amb / med, 1), "subproblems_per_s": round(b / med, 2),
                                  "warmups": warm, "timed_runs": len(runs)})
    finally:
        torch.set_flush_denormal(ftz_default)
        torch.set_num_threads(all_threads)
    best = max(cells, key=lambda c: c["scores_per_s"])
    dep = [c for c in cells if c["batch"] == 1 and c["threads"] == 1 and not c["flush_denormal"]]
    return {"value": best["scores_per_s"], "unit": "scores/s", "cores": best["threads"], "kind": "port",
            "sample": f"best of {len(cells)} cells: {net}, batch {best['batch']} (seed 1234), {best['threads']} thread(s), "
                      f"flush_denormal={best['flush_denormal']}, median of {best['timed_runs']} forwards after {best['warmups']} warm-ups; "
                      f"oracle/gnn_oracle.py; {time.perf_counter() - t_start:.0f}s of CPU work in all",
            "host_cpu": cpu_model_name(), "host_threads": all_threads, "torch_default_flush_denormal": ftz_default,
            "reference_deployment_1core_batch1_ms": dep[0]["ms_per_forward"] if dep else None,
            "cells": cells, "cells_skipped": skipped}


if __name__ == "__main__":
    main()
