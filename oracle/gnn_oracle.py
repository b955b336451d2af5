"""CPU ORACLE for the GNN branching-score forward pass.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / reported CPU baseline.  The product path
(gnn_branching_amd/) never imports it and fails loudly without its HIP library.

What it is: a batch-vectorised torch-CPU restatement of
  graphnet/graph_conv.py   EmbedLayerUpdate.forward :77-388, ComputeFinalScore.forward
                           :442-470, GraphNet.forward :479-483, init_mu :487-496,
                           compute_ratio :499-514
  graphnet/graph_score.py  GraphChoice.decision :21-56 (first-argmax -> [layer, idx])
using the same aten op sequence as the reference (conv2d / conv_transpose2d on a
(B*p, C, H, W) view, addmm for every Linear), without the per-sample Python loops.

Parity pin: tests/golden/*.npz hold inputs and outputs of the REFERENCE ITSELF,
imported unmodified in the authoring container by oracle/make_golden.py
(shipped weights and a seeded non-degenerate weight set; base/wide/deep).
tests/test_oracle_golden.py checks this file against every one of them.
"""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

# state-dict order of the shipped checkpoint; (name, out, in)
GNN_LAYERS = [
    ("EmbedUpdates.update.inp_f", 64, 3), ("EmbedUpdates.update.inp_f_1", 64, 64),
    ("EmbedUpdates.update.inp_b", 64, 2), ("EmbedUpdates.update.inp_b_1", 64, 64),
    ("EmbedUpdates.update.inp_b2", 64, 128), ("EmbedUpdates.update.inp_b2_2", 64, 64),
    ("EmbedUpdates.update.fc1", 64, 7), ("EmbedUpdates.update.fc1_1", 64, 64),
    ("EmbedUpdates.update.fc3", 64, 128), ("EmbedUpdates.update.fc3_2", 64, 64),
    ("EmbedUpdates.update.fc4", 64, 128), ("EmbedUpdates.update.fc4_2", 64, 64),
    ("EmbedUpdates.update.out1", 64, 4), ("EmbedUpdates.update.out2", 64, 128),
    ("EmbedUpdates.update.out3", 64, 64),
    ("EmbedUpdates.update.bc1", 64, 7), ("EmbedUpdates.update.bc1_1", 64, 64),
    ("EmbedUpdates.update.bc1_2", 64, 64), ("EmbedUpdates.update.bc2", 64, 192),
    ("EmbedUpdates.update.bc2_1", 64, 64), ("EmbedUpdates.update.bc3", 64, 128),
    ("EmbedUpdates.update.bc3_1", 64, 64), ("EmbedUpdates.update.bc4", 64, 128),
    ("EmbedUpdates.update.bc4_1", 64, 64),
    ("ComputeFinalScore.fnode", 64, 64), ("ComputeFinalScore.fscore", 1, 64),
]


def random_gnn_state(seed, dtype=np.float32):
    """Seeded NON-DEGENERATE weight set: W ~ N(0, 1/fan_in), b ~ N(0, 0.1^2).

    The shipped checkpoint has 21 all-subnormal tensors (SURVEY.md section 7,
    hard part 1), so parity on it alone does not exercise the forward half-pass.
    """
    rng = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, o, i in GNN_LAYERS:
        sd[name + ".weight"] = (rng.standard_normal((o, i)) / np.sqrt(i)).astype(dtype)
        sd[name + ".bias"] = (0.1 * rng.standard_normal((o,))).astype(dtype)
    return sd


def _lin(sd, name, x):
    # nn.Linear == addmm(bias, x, W^T) (reference graph_conv.py:94 and every other call site)
    return torch.addmm(sd[name + ".bias"], x, sd[name + ".weight"].t())


def compute_ratio(lb, ub):
    """graph_conv.py:499-514, op for op."""
    lower_temp = lb - F.relu(lb)
    upper_temp = F.relu(ub)
    r0 = upper_temp / (upper_temp - lower_temp)
    beta = -1 * lower_temp * r0
    amb = (beta > 0).to(lb.dtype)
    r1 = (1 - 2 * (r0 * amb)) * amb + r0
    return r0, r1, beta, amb


def _fwd_aggregate(layer, mu_src, shape_src, p):
    """nb = A mu_src without bias (graph_conv.py:110-137); returns (B, N_dst, p) and the bias per node."""
    B = mu_src.shape[0]
    if isinstance(layer, nn.Conv2d):
        x = mu_src.permute(0, 2, 1).reshape((B * p,) + tuple(shape_src))          # :112-113
        y = F.conv2d(x, layer.weight.to(x.dtype), None, layer.stride, layer.padding,
                     layer.dilation, layer.groups)                                  # :114
        shape_dst = tuple(y.shape[1:])
        nb = y.reshape(B, p, -1).permute(0, 2, 1)                                   # :118-121
        bias = layer.bias.to(x.dtype).unsqueeze(1).expand(shape_dst[0], shape_dst[1] * shape_dst[2]).reshape(-1)  # :122-123
        return nb, bias, shape_dst
    if isinstance(layer, nn.Linear):
        nb = layer.weight.to(mu_src.dtype) @ mu_src                                 # :131
        return nb, layer.bias.to(mu_src.dtype), (layer.out_features,)
    raise NotImplementedError(type(layer))


def _bwd_aggregate(layer, mu_up, shape_up, p, normalise):
    """nb = A^T mu_up (graph_conv.py:299-326, :361-376); conv case divided by the tap count when ``normalise``."""
    B = mu_up.shape[0]
    if isinstance(layer, nn.Conv2d):
        x = mu_up.permute(0, 2, 1).reshape((B * p,) + tuple(shape_up))            # :302-303
        w = layer.weight.to(x.dtype)
        y = F.conv_transpose2d(x, w, None, layer.stride, layer.padding, 0, layer.groups, layer.dilation)  # :304
        if normalise:                                                               # :306-312
            ones_in = torch.ones(1, 1, shape_up[1], shape_up[2], dtype=x.dtype)
            ones_w = torch.ones(1, 1, w.shape[2], w.shape[3], dtype=x.dtype)
            freq = F.conv_transpose2d(ones_in, ones_w, None, layer.stride, layer.padding, 0, layer.groups, layer.dilation)
            y = y / freq
        return y.reshape(B, p, -1).permute(0, 2, 1)                                 # :316-318
    if isinstance(layer, nn.Linear):
        return layer.weight.to(mu_up.dtype).t().matmul(mu_up)                       # :321
    raise NotImplementedError(type(layer))


def oracle_forward(state, lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_inputs,
                   layers, masks, T=2, p=64, dtype=torch.float32, stages=None, flatten_cls=None):
    """Scores for a batch; same arguments and return value as GraphNet.forward (graph_conv.py:479).

    ``state``: dict name -> array/tensor in the checkpoint's naming.  ``stages``: optional dict that
    receives the embeddings after every half-pass (keys 'r{t}_fwd', 'r{t}_bwd' -> list of (B,N_k,p)).
    """
    sd = {k: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).to(dtype) for k, v in state.items()}
    E = "EmbedUpdates.update."
    lbs = [t.to(dtype) for t in lower_bounds_all]
    ubs = [t.to(dtype) for t in upper_bounds_all]
    duals = [t.to(dtype) for t in dual_vars]
    prim = [torch.as_tensor(t).to(dtype) for t in primals]
    x_lp = primal_inputs.to(dtype)
    fixed = layers["fixed_layers"]
    props = layers["prop_layers"]
    B = lbs[0].shape[0]
    prop_w = torch.stack([pl.weight[0] for pl in props]).to(dtype)      # (B, N_L)
    prop_b = torch.stack([pl.bias[0] for pl in props]).to(dtype)        # (B,)

    mu = [torch.zeros(B, int(np.prod(t.shape[1:])), p, dtype=dtype) for t in lbs]     # init_mu :487-496
    L = len(mu) - 2

    # static structure: for graph layer k (1..L) the incoming linear map and the ReLU's index q
    edges, relu_q = [None], [None]
    pending = None
    for q, l in enumerate(fixed):
        if isinstance(l, (nn.Conv2d, nn.Linear)):
            pending = l
        elif isinstance(l, nn.ReLU):
            edges.append(pending)
            relu_q.append(q)
        elif type(l).__name__ == "Flatten":
            pass
        else:
            raise NotImplementedError(type(l))
    assert len(edges) == L + 1

    for t in range(T):
        if t == 0:                                                                   # :90-95
            inp = torch.stack([lbs[0].reshape(-1), x_lp.reshape(-1), ubs[0].reshape(-1)], 1)
            mu[0] = _lin(sd, E + "inp_f_1", F.relu(_lin(sd, E + "inp_f", inp))).reshape(mu[0].shape)
        # ---- forward sweep :107-192
        for k in range(1, L + 1):
            nb, bias, _ = _fwd_aggregate(edges[k], mu[k - 1], tuple(lbs[k - 1].shape[1:]), p)
            nb = nb.reshape(-1, p)
            l_k, u_k = lbs[k].reshape(-1), ubs[k].reshape(-1)
            r0, r1, beta, amb = compute_ratio(l_k, u_k)                              # :149
            q = relu_q[k]
            feat = torch.stack([beta, l_k, u_k, duals[k - 1][:, 1] - duals[k - 1][:, 2],
                                prim[q - 1], prim[q], bias.repeat(B)], 1)            # :153-159
            relax = _lin(sd, E + "fc1_1", F.relu(_lin(sd, E + "fc1", feat))) * amb.unsqueeze(-1)   # :160-161
            nb_in = torch.cat([nb * r0.unsqueeze(-1), nb * r1.unsqueeze(-1)], 1)    # :169
            e = _lin(sd, E + "fc3_2", F.relu(_lin(sd, E + "fc3", nb_in)))           # :170
            new = _lin(sd, E + "fc4_2", F.relu(_lin(sd, E + "fc4", torch.cat([relax, e], 1))))     # :176-177
            new = new * (r0 != 0).to(dtype).unsqueeze(-1)                            # :178
            if torch.isnan(new).any():
                raise FloatingPointError("mu contains nan")                          # reference :184-186 enters pdb
            mu[k] = new.reshape(mu[k].shape)
        # ---- property node :194-210
        nb = torch.einsum("bn,bnp->bp", prop_w, mu[L])                               # :196
        feat = torch.stack([lbs[L + 1].reshape(-1), ubs[L + 1].reshape(-1), prim[-1], prop_b], 1)  # :202-205
        h = F.relu(_lin(sd, E + "out1", feat))                                       # :206
        mu[L + 1] = _lin(sd, E + "out3", F.relu(_lin(sd, E + "out2", torch.cat([h, nb], 1)))).reshape(mu[L + 1].shape)  # :207-210
        if stages is not None:
            stages[f"r{t}_fwd"] = [m.clone() for m in mu]
        # ---- backward sweep :222-350 (the `ratio` chain :214-216,228,243,356 never feeds an output: omitted)
        for k in range(L, 0, -1):
            l_k, u_k = lbs[k].reshape(-1), ubs[k].reshape(-1)
            r0, r1, beta, amb = compute_ratio(l_k, u_k)                              # :261
            q = relu_q[k]
            e_in = edges[k]
            bias = e_in.bias.to(dtype)
            if isinstance(e_in, nn.Conv2d):                                          # :264-270
                sh = lbs[k].shape
                bias = bias.unsqueeze(1).expand(sh[1], sh[2] * sh[3]).reshape(-1)
            d1, d2 = duals[k - 1][:, 1], duals[k - 1][:, 2]
            feat = torch.stack([l_k, u_k, beta, -d2 + d1, prim[q], prim[q - 1], bias.repeat(B)], 1)  # :273-279
            s = _lin(sd, E + "bc1_2", F.relu(_lin(sd, E + "bc1_1", F.relu(_lin(sd, E + "bc1", feat)))))   # :285
            s2 = torch.cat([s, s * (-d2).unsqueeze(-1), s * d1.unsqueeze(-1)], 1)   # :287-290
            relax = _lin(sd, E + "bc2_1", F.relu(_lin(sd, E + "bc2", s2))) * amb.unsqueeze(-1)      # :291-293
            if k == L:                                                               # next_layer 'prop' :324-326
                nb = prop_w.unsqueeze(-1) * mu[L + 1]                                # (B,N_L,1)*(B,1,p)
            else:
                up = edges[k + 1]
                nb = _bwd_aggregate(up, mu[k + 1], tuple(lbs[k + 1].shape[1:]), p, normalise=True)  # :299-322
            nb = nb.reshape(-1, p)
            nb_in = torch.cat([nb * r0.unsqueeze(-1), nb * r1.unsqueeze(-1)], 1)    # :331-335
            e = _lin(sd, E + "bc3_1", F.relu(_lin(sd, E + "bc3", nb_in)))           # :336
            new = _lin(sd, E + "bc4_1", F.relu(_lin(sd, E + "bc4", torch.cat([relax, e], 1))))     # :344-345
            new = new * (r0 != 0).to(dtype).unsqueeze(-1)                            # :347
            if torch.isnan(new).any():
                raise FloatingPointError("layer_nb_embedding contains nan")         # :339-341
            mu[k] = new.reshape(mu[k].shape)                                         # :349
        # ---- input layer :360-385 (no tap-count division)
        nb = _bwd_aggregate(edges[1], mu[1], tuple(lbs[1].shape[1:]), p, normalise=False).reshape(-1, p)
        inp = torch.stack([lbs[0].reshape(-1), ubs[0].reshape(-1)], 1)               # :380-381
        relax = _lin(sd, E + "inp_b_1", F.relu(_lin(sd, E + "inp_b", inp)))          # :382
        mu[0] = _lin(sd, E + "inp_b2_2", F.relu(_lin(sd, E + "inp_b2", torch.cat([relax, nb], 1)))).reshape(mu[0].shape)
        if stages is not None:
            stages[f"r{t}_bwd"] = [m.clone() for m in mu]

    # ---- scores :442-450
    cat = torch.cat(mu[1:-1], 1)                                                     # (B, R, p)
    scores = []
    for b in range(B):
        sel = cat[b][masks[b].nonzero().view(-1)]
        s = _lin(sd, "ComputeFinalScore.fscore", F.relu(_lin(sd, "ComputeFinalScore.fnode", sel)))
        scores.append(s.view(-1))
    return scores


def decision_from_scores(scores_b, mask_b, relu_sizes):
    """[dec_lay, dec_idx] from one sample's scores (graph_score.py:41-47): FIRST maximal index."""
    if scores_b.numel() == 0:
        raise IndexError("no ambiguous ReLU to branch on")     # reference: torch.max of an empty tensor raises
    choice = int(torch.max(scores_b, 0)[1])
    idx = int(mask_b.nonzero()[choice])
    trans_len = np.cumsum(relu_sizes)
    lay = int(np.nonzero(trans_len > idx)[0][0])
    return [lay, idx if lay == 0 else idx - int(trans_len[lay - 1])]


def padded_scores(scores, masks):
    """(B, R) tensor with -inf where the node is not ambiguous -- the layout the C-ABI returns."""
    out = torch.full(masks.shape, float("-inf"), dtype=scores[0].dtype if scores else torch.float32)
    for b, s in enumerate(scores):
        out[b, masks[b].nonzero().view(-1)] = s
    return out
