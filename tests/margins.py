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
    dirs = [os.path.join(root, "profiles")]
    if os.path.isdir(os.path.join(root, "gpurun_out")) or os.environ.get("GRAFT_REPO_ROOT"):
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        dirs.append(os.path.join(root, "gpurun_out"))
    # the wall-clock records of the BaB-loop stand-in (BASELINE config 5) go to a file of their own
    wall = out["records"].pop("bab_loop_wallclock", None)
    files = [("r06_parity_margins.json", out)] if out["records"] else []
    if wall:
        files.append(("r06_bab_loop_wallclock.json", {k: v for k, v in out.items() if k != "records"} | {
            "note": "BASELINE config 5 stand-in (no Gurobi / CIFAR-10 here): the reference loop's control flow (relu_conv_gnnkwthreshold.py:126-243) on HiGHS-produced "
                    "inputs; GNN scorer calls timed on the MI355X path (GraphChoice.decision, host tensors in, synchronous) and on the CPU twin of the reference "
                    "path (oracle, 1 thread as scripts/bab_mip.sh:3-5 deploys it); tests/test_gpu_bab_trace.py", "records": wall}))
    for d in dirs:
        for name, obj in files:
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(obj, f, indent=1, sort_keys=True)
            except OSError:
                pass
    return os.path.join(dirs[0], files[0][0]) if files else None
