"""``graphnet.graph_conv`` surface of the reference, served by libgnnb.so.

The module tree below exists to carry the PARAMETERS under the reference's
state-dict names -- ``EmbedUpdates.update.<layer>.{weight,bias}`` (25 Linear
layers, reference graph_conv.py:26-74) and ``ComputeFinalScore.{fnode,fscore}``
(:427-432) -- so ``load_state_dict`` of models/cifar_trained_gnn/*.pt works
(graph_score.py:11).  ``GraphNet.forward`` (reference :479-483) hands the batch to
the HIP kernels through the C-ABI; nothing is evaluated with torch ops, and there
is no CPU fallback.
"""
import numpy as np
import torch
from torch import nn

from ..engine import ScorerEngine

# (name, fan-in) in the reference's declaration order (graph_conv.py:36-74); "p" = embedding width
_EMBED_LAYERS = [
    ("inp_f", 3), ("inp_f_1", "p"), ("inp_b", 2), ("inp_b_1", "p"), ("inp_b2", "2p"), ("inp_b2_2", "p"),
    ("fc1", 7), ("fc1_1", "p"), ("fc3", "2p"), ("fc3_2", "p"), ("fc4", "2p"), ("fc4_2", "p"),
    ("out1", 4), ("out2", "2p"), ("out3", "p"),
    ("bc1", 7), ("bc1_1", "p"), ("bc1_2", "p"), ("bc2", "3p"), ("bc2_1", "p"), ("bc3", "2p"), ("bc3_1", "p"),
    ("bc4", "2p"), ("bc4_1", "p"),
]


def _fan_in(spec, p):
    return spec if isinstance(spec, int) else p * int(spec[:-1] or 1)


class EmbedLayerUpdate(nn.Module):
    """Parameter holder for one message-passing update (reference graph_conv.py:22-74)."""

    def __init__(self, p, T):
        super().__init__()
        self.p, self.T = p, T
        for name, spec in _EMBED_LAYERS:
            setattr(self, name, nn.Linear(_fan_in(spec, p), p))


class EmbedUpdates(nn.Module):
    """reference graph_conv.py:394-417"""

    def __init__(self, T, p):
        super().__init__()
        self.T, self.p = T, p
        self.update = EmbedLayerUpdate(p, T)


class ComputeFinalScore(nn.Module):
    """reference graph_conv.py:421-432"""

    def __init__(self, p):
        super().__init__()
        self.p = p
        self.fnode = nn.Linear(p, p)
        self.fscore = nn.Linear(p, 1)


class GraphNet(nn.Module):
    """Drop-in for the reference's ``GraphNet(T, p)`` (graph_conv.py:473-483)."""

    def __init__(self, T, p):
        super().__init__()
        self.T, self.p = T, p
        self.EmbedUpdates = EmbedUpdates(T, p)
        self.ComputeFinalScore = ComputeFinalScore(p)
        self._engine = None
        self._engine_key = None
        self.engine_options = {}      # handle options of the HIP engine (include/gnnb.h gnnb_set_option); set before the first forward

    def _apply(self, fn, *a, **k):            # .cuda() / .to() / .float(): parameters may be replaced
        self._plist = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plist = None
        return super().load_state_dict(*a, **k)

    def engine(self):
        """The HIP engine for the current parameters; re-packed whenever one changed
        (load_state_dict, an optimizer step, .cuda()).  The Parameter objects are listed once: walking the module tree on
        every call cost 0.12 ms of a 0.65 ms decision."""
        if getattr(self, "_plist", None) is None:
            self._plist = list(self.parameters())
        key = tuple((q.data_ptr(), q._version) for q in self._plist)
        if self._engine is None or key != self._engine_key:
            self._engine = ScorerEngine(self.state_dict(), T=self.T, p=self.p, options=self.engine_options)
            self._engine_key = key
        return self._engine

    def load_blob(self, blob):
        """Copy a flat parameter array (checkpoint order, engine.get_weights()) into the module WITHOUT invalidating the
        engine that already holds exactly these parameters (after an online-learning step on the device)."""
        off = 0
        with torch.no_grad():
            for q in self.state_dict().values():
                n = q.numel()
                q.copy_(torch.from_numpy(np.asarray(blob[off:off + n], dtype=np.float32)).reshape(q.shape))
                off += n
        if self._engine is not None:
            self._plist = list(self.parameters())
            self._engine_key = tuple((q.data_ptr(), q._version) for q in self._plist)

    def forward_device(self, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks):
        """Batched forward; returns the device-resident ForwardResult (padded scores, decisions,
        status) without any host synchronisation."""
        return self.engine().forward(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks)

    def forward(self, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks):
        """Same arguments and return value as the reference (graph_conv.py:479): a list with one
        1-D tensor per subproblem holding the scores of its ambiguous ReLUs in flat ReLU order."""
        res = self.forward_device(lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs, layers, masks)
        return res.check().ragged()
