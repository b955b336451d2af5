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
    assert lib.gnnb_abi_version() == 2
    n = lib.gnnb_profile_classes()
    names = [lib.gnnb_profile_class_name(i).decode() for i in range(n)]
    assert "k_node_update" in names and "k_conv_fwd" in names and len(set(names)) == n


def test_option_table_is_the_documented_one(lib):
    """Handle options go through the C-ABI (gnnb_set_option), not the environment: the table the library exports == the names the Python
    host knows == the table of include/gnnb.h and INTEGRATION.md; ten at most; unknown names and null handles are refused without a GPU."""
    names = [lib.gnnb_option_name(i).decode() for i in range(lib.gnnb_option_count())]
    assert sorted(names) == sorted(_lib.OPTIONS) and len(names) <= 10, names
    header = open(os.path.join(ROOT, "include", "gnnb.h")).read()
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in names:
        assert f'"{n}"' in header, n
        assert f"`{n}`" in integ, n
    assert lib.gnnb_set_option(None, b"fuse", 0) != 0
    assert lib.gnnb_option_name(99) == b""


def test_library_and_package_leave_the_environment_alone():
    """The shipped library reads no environment variable (development builds, -DGNNB_DEV, do); the package never WRITES one: options reach a
    handle through gnnb_set_option (ScorerEngine(options=...)), the GNNB_* names of _lib.OPTION_ENV are read once per engine for the tests."""
    csrc = os.path.join(ROOT, "gnn_branching_amd", "csrc")
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".h")):
            continue
        depth = 0                                      # nesting depth of #ifdef GNNB_DEV blocks
        stack = []
        for line in open(os.path.join(csrc, f)):
            t = line.strip()
            if t.startswith(("#if", "#ifdef", "#ifndef")):
                stack.append("GNNB_DEV" in t and not t.startswith("#ifndef"))
                depth += stack[-1]
            elif t.startswith("#else") and stack and stack[-1]:
                stack[-1] = False
                depth -= 1
            elif t.startswith("#endif") and stack:
                depth -= stack.pop()
            elif "getenv(" in t and not t.startswith("//"):
                assert depth > 0, f"{f}: getenv outside #ifdef GNNB_DEV: {t}"
    pkg = os.path.join(ROOT, "gnn_branching_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"os\.environ\[[^\]]+\]\s*=|os\.environ\.(setdefault|update|pop)\(|os\.putenv|del os\.environ", src), os.path.join(d, f)
    assert _lib.options_from_env({"GNNB_NO_TOP": "1", "GNNB_TOP_SPLIT": "3", "GNNB_BF3": "0", "OTHER": "7"}) == {"top": 0, "top_split": 2, "bf3": 0}
    assert _lib.options_from_env({}) == {}


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


def test_status_word_maps_to_the_reference_s_failure_modes():
    """gnnb_forward's status word: bit 0 = a NaN embedding (the reference drops into pdb there, graph_conv.py:184-186, :339-341) ->
    FloatingPointError; bit 1 = a wait inside a kernel (the fused half-pass ring, k_top's workgroup split) hit its iteration cap -> RuntimeError (results invalid);
    the words of several chunks are OR-ed; 0 raises nothing."""
    import pytest
    import torch
    from gnn_branching_amd import engine
    assert engine._or_reduce(torch.tensor([0, 0], dtype=torch.int32)) == 0
    assert engine._or_reduce(torch.tensor([1, 2], dtype=torch.int32)) == 3
    engine._raise_for_status(0)
    with pytest.raises(FloatingPointError, match="nan"):
        engine._raise_for_status(1)
    with pytest.raises(RuntimeError, match="iteration cap"):
        engine._raise_for_status(2)
    with pytest.raises(RuntimeError):                 # a timed-out wait outranks the NaN check: its rows are garbage anyway
        engine._raise_for_status(3)
    with pytest.raises(RuntimeError, match="record image"):      # bit 2: gnnb_scatter_amb_records refused a foreign / corrupt image
        engine._raise_for_status(4)
    res = engine.ForwardResult(None, None, torch.tensor([0, 1], dtype=torch.int32), None)
    with pytest.raises(FloatingPointError):
        res.check()


def test_library_carries_the_hash_of_its_sources(lib, tmp_path, monkeypatch):
    """`*.so` is git-ignored yet ships with the snapshot, so mtimes say nothing: the build id compiled into the library is the
    hash of the sources it came from; `needs_build()` compares it with the tree's, and a binary with another id is stale."""
    want = _lib.source_hash()
    assert re.fullmatch(r"[0-9a-f]{32}", want)
    assert lib.gnnb_build_id().decode() == want == _lib.library_build_id()
    assert not _lib.needs_build()
    # a binary from other sources (here: the same bytes with a patched id), however new its timestamp, is not current
    blob = open(_lib.LIB_PATH, "rb").read()
    i = blob.find(_lib.BUILD_ID_MARK) + len(_lib.BUILD_ID_MARK)
    stale = tmp_path / "libgnnb.so"
    stale.write_bytes(blob[:i] + b"0" * 32 + blob[i + 32:])
    assert _lib.library_build_id(str(stale)) == "0" * 32
    monkeypatch.setattr(_lib, "LIB_PATH", str(stale))
    assert _lib.needs_build()
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    assert _lib.needs_build()


def test_graft_entry_build_runs():
    """The driver's "does it build" step (__graft_entry__.build) must pass on the tree as it is -- its ABI assertions included."""
    import __graft_entry__ as g
    g.build()


def test_tile_sample_magic_division_is_exact():
    """gnnb_dev.h tile_sample: sample = tile / TPS as the high word of tile * (floor(2^32 / TPS) + 1) plus one compare (the magic number comes from
    gnnb.hip to_dtm).  Emulated here over every divisor the tile tables allow and tile indices up to 2^32 - 1: the quotient is exact."""
    import numpy as np
    rng = np.random.RandomState(0)
    for d in list(range(2, 4097)) + [65535, 1 << 20, (1 << 31) - 1]:
        magic = ((1 << 32) // d + 1) & 0xFFFFFFFF
        n = np.concatenate([np.arange(0, min(4 * d + 2, 4096), dtype=np.uint64), rng.randint(0, 1 << 32, size=64, dtype=np.uint64),
                            np.array([(1 << 32) - 1, (1 << 31), d * ((1 << 32) // d) - 1, d * ((1 << 32) // d)], dtype=np.uint64) & np.uint64(0xFFFFFFFF)])
        q = (n * np.uint64(magic)) >> np.uint64(32)
        q = q - ((q * np.uint64(d)) > n).astype(np.uint64)
        assert np.array_equal(q, n // np.uint64(d)), d
