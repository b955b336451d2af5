"""Test infrastructure (not product code): extracts the property table of the reference's experiment file cifar_exp/base_easy.pkl -- the
rows `experiments/bab_mip.py --bab_gnn` iterates over (bab_mip.py:91-120: image index, eps, target class) together with what the
reference recorded for its own GNN + KW-threshold run on them (BBran_gnnkwT branches, BTime_gnnkwT seconds, BSAT_gnnkwT) -- into the
small fixture tests/golden/base_easy_props.npz.  DATA only (no reference source); runs in the authoring container where /root/reference
exists:   python oracle/make_golden_props.py
The CIFAR-10 images the indices point at are not available offline, so the tests use the (Eps, prop) pairs with seeded stand-in inputs."""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/cifar_exp/base_easy.pkl"
d = pd.read_pickle(src)
n = 32
out = {
    "source": np.array("cifar_exp/base_easy.pkl (first %d of %d rows)" % (n, len(d))),
    "Idx": d["Idx"].to_numpy()[:n].astype(np.int64),
    "Eps": d["Eps"].to_numpy()[:n].astype(np.float64),
    "prop": d["prop"].to_numpy()[:n].astype(np.int64),
    "BBran_gnnkwT": d["BBran_gnnkwT"].to_numpy()[:n].astype(np.float64),
    "BTime_gnnkwT": d["BTime_gnnkwT"].to_numpy()[:n].astype(np.float64),
    "BSAT_gnnkwT": np.array([str(v) for v in d["BSAT_gnnkwT"].to_list()[:n]]),
    "all_rows_mean_BBran_gnnkwT": np.array(float(pd.to_numeric(d["BBran_gnnkwT"], errors="coerce").mean())),
    "all_rows_mean_BTime_gnnkwT": np.array(float(pd.to_numeric(d["BTime_gnnkwT"], errors="coerce").mean())),
}
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "base_easy_props.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
print("first rows:", list(zip(out["Idx"][:4], out["Eps"][:4], out["prop"][:4], out["BBran_gnnkwT"][:4], out["BTime_gnnkwT"][:4])))
print("mean branches / seconds over all rows:", out["all_rows_mean_BBran_gnnkwT"], out["all_rows_mean_BTime_gnnkwT"])
