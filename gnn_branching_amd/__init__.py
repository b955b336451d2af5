"""MI355X-native GNN branching scorer (drop-in for oval-group/GNN_branching's graphnet path)."""
import importlib
import sys
import types

__version__ = "0.2.0"

# import path of the reference -> module of this package that takes its place.  Only the modules of the hot path are
# aliased; everything else of the reference's `plnn` / `graphnet` packages (plnn.relu_conv_gnnkwthreshold,
# plnn.branch_and_bound, plnn.conv_kwinter_gen, ...) keeps resolving to the reference's own files.
DROPIN_MODULES = {
    "graphnet.graph_score": "gnn_branching_amd.graphnet.graph_score",      # GraphChoice (relu_conv_gnnkwthreshold.py:7)
    "graphnet.graph_conv": "gnn_branching_amd.graphnet.graph_conv",        # GraphNet
    "plnn.modules": "gnn_branching_amd.plnn.modules",                      # Flatten (graph_conv.py:11)
}
DROPIN_ONLINE = {"graphnet.graph_score_online": "gnn_branching_amd.graphnet.graph_score_online"}   # relu_conv_online.py:7
# names other modules may already have bound with `from <alias> import <name>` before install_dropin() ran
_REBIND = {"graphnet.graph_score": ("GraphChoice",), "graphnet.graph_conv": ("GraphNet",), "plnn.modules": ("Flatten",),
           "graphnet.graph_score_online": ("GraphChoice",)}


def _parent_package(name):
    """The reference's own package `name` if it can be imported (it stays what it is: its other submodules must keep
    working), else an empty stub package so that ``import name.sub`` finds the aliased submodules."""
    mod = sys.modules.get(name)
    if mod is not None:
        return mod
    try:
        return importlib.import_module(name)
    except ModuleNotFoundError as e:
        if e.name != name:
            raise
    mod = types.ModuleType(name)
    mod.__path__ = []                       # a package without files of its own
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


def install_dropin(online=False, babsr=False):
    """Make ``graphnet.graph_score`` / ``graphnet.graph_conv`` / ``plnn.modules`` resolve to this package, so the
    reference's BaB driver (``from graphnet.graph_score import GraphChoice``, plnn/relu_conv_gnnkwthreshold.py:7) picks up
    the MI355X scorer without edits.  The reference's ``plnn`` and ``graphnet`` PACKAGES are left alone: only the three
    submodules are entered into ``sys.modules`` (and set as attributes of their parent package), so
    ``from plnn.relu_conv_gnnkwthreshold import relu_gnn``, ``import plnn.branch_and_bound`` etc. still import the
    reference's files.

    online=True also aliases ``graphnet.graph_score_online`` (the online-learning GraphChoice, relu_conv_online.py:7);
    babsr=True replaces ``choose_node_conv`` inside the reference's ``plnn.kw_score_conv`` (its other functions stay).

    Works before or after the reference's packages were imported; modules that already executed
    ``from graphnet.graph_score import GraphChoice`` against the reference's class are re-pointed."""
    aliases = dict(DROPIN_MODULES)
    if online:
        aliases.update(DROPIN_ONLINE)
    installed = {}
    for alias, target in aliases.items():
        parent_name, leaf = alias.rsplit(".", 1)
        parent = _parent_package(parent_name)
        mod = importlib.import_module(target)
        old = sys.modules.get(alias)
        sys.modules[alias] = mod
        setattr(parent, leaf, mod)
        installed[alias] = mod
        if old is not None and old is not mod:             # late install: re-point names imported from the replaced module
            for name in _REBIND.get(alias, ()):
                old_obj, new_obj = getattr(old, name, None), getattr(mod, name)
                if old_obj is None:
                    continue
                for m in list(sys.modules.values()):
                    if m is None or m is mod or m is old:
                        continue
                    if getattr(m, "__dict__", {}).get(name) is old_obj:
                        setattr(m, name, new_obj)
    if babsr:
        from .plnn import kw_score_conv as ours
        try:
            theirs = importlib.import_module("plnn.kw_score_conv")
        except ModuleNotFoundError:
            _parent_package("plnn")
            sys.modules["plnn.kw_score_conv"] = ours
            setattr(sys.modules["plnn"], "kw_score_conv", ours)
            theirs = ours
        if theirs is not ours:
            old_fn = getattr(theirs, "choose_node_conv", None)
            theirs.choose_node_conv = ours.choose_node_conv
            for m in list(sys.modules.values()):
                if m is not None and old_fn is not None and getattr(m, "__dict__", {}).get("choose_node_conv") is old_fn:
                    m.choose_node_conv = ours.choose_node_conv
        installed["plnn.kw_score_conv.choose_node_conv"] = ours.choose_node_conv
    return installed
