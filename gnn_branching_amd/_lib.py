"""ctypes binding of libgnnb.so (C-ABI in include/gnnb.h) and its in-tree build.

No HIP call is made at import or load time: the BaB harness forks one child per
property and creates the GPU context inside the child (reference
experiments/bab_mip.py:244-249), so device work starts at the first forward.
There is NO CPU fallback: if the library is missing or no MI355X is visible the
product path raises.
"""
import ctypes as C
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("GNNB_LIB", os.path.join(CSRC, "libgnnb.so"))     # GNNB_LIB: dev override (ablation builds)
SOURCES = ["gnnb.hip", "gnnb_dev.h", "gnnb_k_mlp.h", "gnnb_k_gather.h", "gnnb_k_fusedq.h", "gnnb_k_edges.h", "gnnb_k_misc.h", "gnnb_pack.h", "gnnb_train.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread"]

GNNB_CONV, GNNB_LINEAR, GNNB_RELU, GNNB_FLATTEN = 0, 1, 2, 3


class LayerDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("c_in", C.c_int32), ("c_out", C.c_int32), ("kh", C.c_int32),
                ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("n_in", C.c_int32),
                ("n_out", C.c_int32), ("weight", C.c_void_p), ("bias", C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [("lb", C.POINTER(C.c_void_p)), ("ub", C.POINTER(C.c_void_p)),
                ("dual", C.POINTER(C.c_void_p)), ("primal", C.POINTER(C.c_void_p)),
                ("x_lp", C.c_void_p), ("prop_w", C.c_void_p), ("prop_b", C.c_void_p), ("mask", C.c_void_p),
                ("n_graph", C.c_int32), ("n_relu", C.c_int32), ("n_primal", C.c_int32)]


# every symbol include/gnnb.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("gnnb_create", C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    ("gnnb_bind_network", C.c_int, [C.c_void_p, C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int, C.c_int]),
    ("gnnb_graph_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("gnnb_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int]),
    ("gnnb_forward", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    ("gnnb_forward_host", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("gnnb_amb_records_bytes", C.c_size_t, [C.c_void_p, C.c_int]),
    ("gnnb_pack_amb_records", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("gnnb_scatter_amb_records", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                                           C.c_void_p]),
    ("gnnb_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("gnnb_get_option", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    ("gnnb_option_count", C.c_int, []),
    ("gnnb_option_name", C.c_char_p, [C.c_int]),
    ("gnnb_babsr", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p,
                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("gnnb_mu_projection", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("gnnb_destroy", C.c_int, [C.c_void_p]),
    ("gnnb_get_weights", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("gnnb_set_weights", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("gnnb_online_create", C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    ("gnnb_online_step", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p]),
    ("gnnb_online_grad", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("gnnb_last_error", C.c_char_p, []),
    ("gnnb_abi_version", C.c_int, []),
    ("gnnb_build_id", C.c_char_p, []),
    ("gnnb_describe", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    ("gnnb_mu_location", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("gnnb_set_halfpass_limit", C.c_int, [C.c_void_p, C.c_int]),
    ("gnnb_debug_occupy", C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_void_p]),
    ("gnnb_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("gnnb_profile_classes", C.c_int, []),
    ("gnnb_profile_class_name", C.c_char_p, [C.c_int]),
    ("gnnb_profile_read", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int, C.c_int]),
    ("gnnb_profile_trace", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]),
]


# Handle options (gnnb_set_option) and the environment names the PYTHON host accepts for them.  libgnnb.so reads no environment
# variable; `ScorerEngine(options=...)` is the interface.  The environment names exist for the parity tests and bench.py's side legs,
# which select an implementation per test with monkeypatch.setenv: they are read once, when an engine is created, and go through
# gnnb_set_option like any other option (never written by this package).  value = (option, transform of the variable's integer).
OPTION_ENV = {
    "GNNB_BF3": ("bf3", lambda v: int(v != 0)),
    "GNNB_FUSE": ("fuse", lambda v: int(v != 0)),
    "GNNB_NO_TOP": ("top", lambda v: int(v == 0)),
    "GNNB_NO_GATHER": ("gather", lambda v: int(v == 0)),
    "GNNB_NO_EMBED_FUSE": ("embed_fuse", lambda v: int(v == 0)),
    "GNNB_NO_DENSE_LDS": ("dense_lds", lambda v: int(v == 0)),
    "GNNB_TAIL_MAX_B": ("tail_max_b", lambda v: max(0, min(v, 1 << 30))),
    "GNNB_TOP_SPLIT": ("top_split", lambda v: 4 if v >= 4 else (2 if v >= 2 else 1)),
    "GNNB_TOP_FUSE_UPD": ("top_fuse_upd", lambda v: int(v != 0)),
    "GNNB_CLSPRE_MAX_B": ("clspre_max_b", lambda v: max(0, min(v, 1 << 30))),
}
OPTIONS = tuple(sorted({o for o, _ in OPTION_ENV.values()}))


def options_from_env(env=None):
    """{option: value} for the OPTION_ENV variables that are set (read-only view of the environment)."""
    env = os.environ if env is None else env
    out = {}
    for name, (opt, f) in OPTION_ENV.items():
        v = env.get(name)
        if v is not None and v.strip() != "":
            out[opt] = f(int(v))
    return out


BUILD_ID_MARK = b"GNNB_BUILD_ID:"


def source_hash():
    """sha256 (first 32 hex digits) over the compiler flags and the text of every source the library is built from.  It is
    compiled into the library (`gnnb_build_id()`), so a binary always says which sources it came from."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for name in SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    with open(os.path.join(CSRC, "..", "..", "include", "gnnb.h"), "rb") as f:
        h.update(b"include/gnnb.h\0" + f.read())
    return h.hexdigest()[:32]


def library_build_id(path=None):
    """The source hash a built library carries (read from the file, without loading it); None if it carries none."""
    try:
        with open(path or LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(BUILD_ID_MARK)
    if i < 0:
        return None
    return blob[i + len(BUILD_ID_MARK):i + len(BUILD_ID_MARK) + 32].decode("ascii", "replace")


def needs_build():
    """The library is current iff it carries the hash of the sources in the tree -- not its mtime: `*.so` is git-ignored but
    ships to the GPU box with the snapshot, and a stale binary whose timestamp happens to be newer must not be reused."""
    return library_build_id() != source_hash()


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> csrc/libgnnb.so (cross-compiles without a GPU).

    Safe to call from several processes at once (one rank per GPU under torchrun): the build runs under a file lock, into a
    temporary file that replaces the library atomically, and whoever gets the lock second finds the library up to date."""
    if "GNNB_LIB" in os.environ or (not force and not needs_build()):
        return LIB_PATH
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            tmp = LIB_PATH + f".tmp{os.getpid()}"
            cmd = ["hipcc"] + HIPCC_FLAGS + [f'-DGNNB_SRC_HASH="{source_hash()}"', "-o", tmp, os.path.join(CSRC, "gnnb.hip")]
            if verbose:
                print(" ".join(cmd).replace(tmp, LIB_PATH), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("building libgnnb.so failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_lib = None


def load():
    """dlopen the library and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the branching scorer)")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so): it must be the one already in
        # the process when libgnnb.so is dlopened, or two runtimes fight over the device ("no
        # ROCm-capable device is detected").  Importing torch does not initialise the GPU.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        if "GNNB_LIB" not in os.environ and os.path.exists(os.path.join(CSRC, "gnnb.hip")):
            if not hasattr(lib, "gnnb_build_id"):      # a library from before the source hash was compiled in
                raise RuntimeError(f"{LIB_PATH} carries no build id (built from older sources): run "
                                   "`python -c 'import __graft_entry__ as g; g.build()'`")
            lib.gnnb_build_id.restype = C.c_char_p
            have, want = lib.gnnb_build_id().decode("ascii", "replace"), source_hash()
            if have != want:
                raise RuntimeError(f"{LIB_PATH} was built from other sources (library {have}, tree {want}): run "
                                   "`python -c 'import __graft_entry__ as g; g.build()'`")
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().gnnb_last_error().decode(errors="replace")
        print(f"[gnn_branching_amd] {what} failed ({rc}): {msg}", file=sys.stderr)
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
