"""dev helper (GPU box): host time of gnnb_set_weights (pack rebuild + upload), the tail of every online step."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gnn_branching_amd.engine import ScorerEngine
from tests.common import shipped_state
eng = ScorerEngine(shipped_state())
w = eng.get_weights()
eng.set_weights(w)
t0 = time.perf_counter()
for _ in range(10):
    eng.set_weights(w)
print(f"gnnb_set_weights: {1e2 * (time.perf_counter() - t0):.2f} ms")
