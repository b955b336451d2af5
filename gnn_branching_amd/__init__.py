"""MI355X-native GNN branching scorer (drop-in for oval-group/GNN_branching's graphnet path)."""
import sys

__version__ = "0.1.0"


def install_dropin():
    """Make ``import graphnet.graph_score`` / ``graphnet.graph_conv`` / ``plnn.modules`` resolve to
    this package, so the reference's BaB driver (plnn/relu_conv_gnnkwthreshold.py:7) picks up the
    MI355X scorer without edits.  Call before the reference modules are imported."""
    from . import graphnet as g
    from .graphnet import graph_conv, graph_score
    from . import plnn as p
    from .plnn import modules
    for name, mod in (("graphnet", g), ("graphnet.graph_conv", graph_conv), ("graphnet.graph_score", graph_score),
                      ("plnn.modules", modules)):
        sys.modules[name] = mod
    if "plnn" not in sys.modules:
        sys.modules["plnn"] = p
    else:
        sys.modules["plnn"].modules = modules
