"""Parity margins on record: the GPU parity tests report, per configuration, how far the HIP scores were from the oracle, how large the
scores were and how close the closest decision was; tests/conftest.py writes what was collected to profiles/r06_parity_margins.json
when the session ends (`pytest -q` prints dots: the numbers behind a green run would otherwise be lost).  On a gpurun box the file is
also copied to gpurun_out/, which is what travels back."""
import json
import os
import time

_RECORDS = {}


def record(section, key, **fields):
    """Keep the WORST of repeated reports under (section, key): fields named `worst_*` / `max_*` take the maximum, `min_*` the minimum,
    `n_*` add up, everything else is overwritten."""
    cur = _RECORDS.setdefault(section, {}).setdefault(key, {})
    for k, v in fields.items():
        if isinstance(v, (int, float)) and k in cur and isinstance(cur[k], (int, float)):
            if k.startswith(("worst_", "max_")):
                v = max(cur[k], v)
            elif k.startswith("min_"):
                v = min(cur[k], v)
            elif k.startswith("n_"):
                v = cur[k] + v
        cur[k] = v


def dump(root):
    if not _RECORDS:
        return None
    out = {"written": time.strftime("%Y-%m-%d %H:%M:%S"), "note": "worst |HIP - oracle| per configuration as measured by this pytest session "
           "(tests/margins.py); tolerances: tests/common.py (shipped checkpoint 1e-4 absolute = north_star, seeded random set 5e-6)",
           "records": _RECORDS}
    try:
        import torch
        if torch.cuda.is_available():
            out["device"] = torch.cuda.get_device_name(0)
        from gnn_branching_amd import _lib
        out["library_build_id"] = _lib.library_build_id()
    except Exception:      # noqa: BLE001
        pass
    paths = [os.path.join(root, "profiles", "r06_parity_margins.json")]
    if os.path.isdir(os.path.join(root, "gpurun_out")) or os.environ.get("GRAFT_REPO_ROOT"):
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        paths.append(os.path.join(root, "gpurun_out", "r06_parity_margins.json"))
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1, sort_keys=True)
        except OSError:
            pass
    return paths[0]
