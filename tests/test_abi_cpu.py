"""The C-ABI library builds for gfx950, loads without a GPU, and exports every symbol of include/gnnb.h."""
import os
import re

import pytest

from gnn_branching_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    _lib.build_library()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "gnnb.h")).read()
    declared = set(re.findall(r"\b(gnnb_[a-z_]+)\s*\(", header))
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name


def test_no_device_needed_for_metadata(lib):
    assert lib.gnnb_abi_version() == 1
    n = lib.gnnb_profile_classes()
    names = [lib.gnnb_profile_class_name(i).decode() for i in range(n)]
    assert "k_node_update" in names and "k_conv_fwd" in names and len(set(names)) == n


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gnn_branching_amd.graphnet.graph_conv import GraphNet
    from gnn_branching_amd import synth
    batch = synth.make_batch("cifar_base_kw", 1, seed=0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        GraphNet(2, 64)(*batch.forward_args())


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gnn_branching_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(d, f)
