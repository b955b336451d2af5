"""CPU: the accounting helpers of bench.py (no GPU, no timing): which counter summaries belong to which profile class, the
SURVEY section 8(d) byte count of message passing, and that the committed counter summaries cover every half-pass kernel class of the
three bench configurations (otherwise `roofline_message_passing.frac_counter` silently comes out as null)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.abspath(bench.__file__))


def test_pmc_rows_match_kernel_templates_to_profile_classes():
    row = {"hbm_bytes_per_launch": 1.0, "launches_sampled": 2}
    pmc = {"k_gather": row, "k_gather16": row, "k_gather_scored": row, "k_gather_update_q": row, "k_gather_input_update": row,
           "k_node_update": row, "k_top": row, "k_gather_no_counters": {"launches_sampled": 1}}
    assert len(bench.pmc_rows(pmc, "k_gather")) == 3                 # k_gather, k_gather16, k_gather_scored -- not the fused or input kernels
    assert len(bench.pmc_rows(pmc, "k_gather_update")) == 1          # the kernel is k_gather_update_q
    assert len(bench.pmc_rows(pmc, "k_gather_input_update")) == 1
    assert len(bench.pmc_rows(pmc, "k_node_update")) == 1
    assert bench.pmc_rows(pmc, "k_conv_fwd") == []


def test_message_passing_bytes_follow_survey_8d():
    # cifar_base_kw graph layers: input 3072, ReLU layers 2048 / 1024 / 100, property node 1; p = 64 floats per row
    sizes = [3072, 2048, 1024, 100, 1]
    B, T = 2, 2
    fwd = (3072 + 2048) + (2048 + 1024) + (1024 + 100) + (100 + 1)
    bwd = (1024 + 2048) + (100 + 1024) + (1 + 100)
    inp = 2048 + 3072
    want = 4.0 * 64 * B * (T * (fwd + bwd) + (T - 1) * inp)
    assert bench.message_passing_bytes(sizes, B, T) == want
    assert want / B == 8332288.0                                     # the per-subproblem figure DESIGN.md section 5 quotes


def test_committed_counter_summaries_cover_the_half_pass_kernels():
    # (round 4: the restricted last half-pass runs inside k_scored_tail on the default path: no stand-alone k_gather / k_node_update launch)
    mp_names = ("k_gather", "k_gather_update", "k_gather_input_update", "k_top", "k_node_update", "k_scored_tail")
    for net, B in (("cifar_base_kw", 256), ("cifar_wide_kw", 256), ("cifar_deep_kw", 128)):
        path = os.path.join(ROOT, "profiles", f"pmc_latest_{net}_B{B}.json")
        assert os.path.exists(path), path
        pmc = json.load(open(path))
        for cls in mp_names:
            rows = bench.pmc_rows(pmc, cls)
            if cls in ("k_gather", "k_node_update") and not rows:
                continue                                             # a configuration may not launch a stand-alone gather / node update at all
            assert rows and all(r["hbm_bytes_per_launch"] > 0 for r in rows), (net, cls)
