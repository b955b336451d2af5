"""Mirror of the reference's ``graphnet`` package: same class names, signatures and
state-dict keys, backed by the MI355X HIP library (no PyTorch math on the hot path)."""
