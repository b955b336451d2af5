"""dev helper (GPU box): latency of one online-learning step (gnnb_online_step) vs. the autograd CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from gnn_branching_amd import synth
from gnn_branching_amd.engine import ScorerEngine
from tests.common import shipped_state

net = sys.argv[1] if len(sys.argv) > 1 else "cifar_base_kw"
for B in [int(v) for v in os.environ.get("ONLINE_B", "1,8").split(",")]:
    batch = synth.make_batch(net, B, seed=1234)
    eng = ScorerEngine(shipped_state())
    eng.online_create()
    kws = [int(batch.masks[b].nonzero().view(-1)[3]) for b in range(B)]
    imps = [0.1] * B
    args = batch.forward_args()
    dev = eng._marshal(*args)          # inputs resident
    args_dev = (dev[1], dev[2], dev[3], dev[4], dev[5], args[5], dev[6].view(B, -1))
    for _ in range(3):
        eng.online_step(args_dev, kws, imps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.online_step(args_dev, kws, imps)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        eng.online_step(args_dev, kws, imps, apply=False)
    torch.cuda.synchronize()
    t_grad = (time.perf_counter() - t0) / n
    line = f"{net} B={B}: online step {1e3 * t_all:.2f} ms (forward+backward only {1e3 * t_grad:.2f} ms)"
    if "--cpu" in sys.argv:
        from oracle.online_oracle import OnlineOracle
        torch.set_num_threads(int(os.environ.get("CPU_THREADS", "16")))
        o = OnlineOracle(shipped_state())
        o.step(args, kws, imps)
        t0 = time.perf_counter()
        for _ in range(3):
            o.step(args, kws, imps)
        line += f"; CPU oracle (torch autograd, {torch.get_num_threads()} threads) {1e3 * (time.perf_counter() - t0) / 3:.1f} ms"
    print(line, flush=True)
