"""``plnn.modules`` boundary types.

The GNN forward dispatches on ``type(layer) is Flatten`` for the verified
network's layer list (reference graphnet/graph_conv.py:188, :355), so a
``Flatten`` marker module must exist under this import path
(reference plnn/modules.py:4-6).
"""
from torch import nn


class Flatten(nn.Module):
    """(B, C, H, W) -> (B, C*H*W); a shape-only marker between conv and linear layers."""

    def forward(self, x):
        return x.reshape(x.shape[0], -1)
