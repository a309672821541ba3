"""CPU oracle for GWEN's GCNConv-stack forward  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Nothing under ``gwen_amd/`` imports it; the product path has no CPU fallback.

PARITY UNPINNED.  The arithmetic of the reference's hot path lives in an un-vendored third-party
dependency, ``torch-geometric==2.3.1`` (pin: /root/reference/requirements/environment.yml:552),
which is absent from the reference tree and not installable here; the reference's own tests
(/root/reference/tests/test_gwen/test_models.py:19,36) mock the GNN layers out and hold no numeric
golden vector.  This file therefore *restates* the published GCN propagation rule
``X' = D^-1/2 (A+I) D^-1/2 X W^T + b`` with the option values the reference's call sites fix, and is
anchored by closed-form known-answer tests (tests/test_oracle_kat.py) and by an independent plain-C
restatement (oracle/gcn_ref.c) that it is cross-checked against.

What the reference fixes (all file:line relative to /root/reference):
  * constructors ``GCNConv(in, out)`` with no kwargs: src/gwen/models_gnn.py:118-130, :172-184
    => improved=False, cached=False, add_self_loops=True, normalize=True, bias=True,
       flow="source_to_target", aggr="add".
  * calls ``conv(x, edge_index)`` with no edge_weight: src/gwen/models_gnn.py:147-149, :204-206.
  * composition conv1,conv2,conv3 (ReLU each) then upconv3,upconv4 (ReLU) and upconv5 (no
    activation): src/gwen/models_gnn.py:135-157, :189-212, :241-258, :292-303.
  * edge_index is int64 [2,E], row 0 = source j, row 1 = target i (PyG convention), produced by
    ``erdos_renyi_graph(num_members, edge_prob=1)``: src/gwen/utils.py:176.

The op sequence below deliberately mirrors what PyG executes on CPU (index_select -> mul ->
index_add_ in edge order, self-loops appended last), because that is the "reference PyTorch CPU
path" whose time is reported as ``cpu_baseline`` and whose summation order the HIP kernels reproduce.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor, nn


# --------------------------------------------------------------------------------------------
# gcn_norm  (invoked inside every GCNConv.forward; call sites models_gnn.py:147-149,204-206)
# --------------------------------------------------------------------------------------------
def add_remaining_self_loops(
    edge_index: Tensor, edge_weight: Optional[Tensor], fill_value: float, num_nodes: int
) -> Tuple[Tensor, Optional[Tensor]]:
    """Drop explicit self-loops, append one (k,k) per node k = 0..N-1 *after* the kept edges.

    With weights: a pre-existing loop keeps its own weight (last duplicate wins), every other node's
    loop gets ``fill_value``.  Without weights the result carries no weights (filled with ones later).
    """
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    if edge_weight is not None:
        loop_w = edge_weight.new_full((num_nodes,), fill_value)
        inv = ~keep
        # sequential assignment => last duplicate wins (CPU semantics)
        for n, w in zip(row[inv].tolist(), edge_weight[inv].tolist()):
            loop_w[n] = w
        edge_weight = torch.cat([edge_weight[keep], loop_w])
    edge_index = torch.cat([edge_index[:, keep], torch.stack([loop, loop])], dim=1)
    return edge_index, edge_weight


def gcn_norm(
    edge_index: Tensor,
    edge_weight: Optional[Tensor],
    num_nodes: int,
    improved: bool = False,
    add_self_loops: bool = True,
    dtype: torch.dtype = torch.float32,
) -> Tuple[Tensor, Tensor]:
    """Symmetric normalisation  w~_e = d^-1/2[src] * w_e * d^-1/2[dst],  d = in-degree incl. loop.

    As in torch-geometric 2.3.x the self-loops are completed BEFORE unit weights are materialised, so
    without explicit edge weights every loop weighs 1 and ``improved`` (fill 2) only acts on weighted
    graphs (restated from the library's behaviour; the reference never sets ``improved``)."""
    fill = 2.0 if improved else 1.0
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill, num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    else:
        edge_weight = edge_weight.to(dtype)
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(num_nodes, dtype=dtype).index_add_(0, col, edge_weight)  # over TARGETS
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    w = dis[row] * edge_weight * dis[col]
    return edge_index, w


# --------------------------------------------------------------------------------------------
# one GCNConv layer  (lin -> propagate -> + bias)
# --------------------------------------------------------------------------------------------
def propagate(h: Tensor, edge_index: Tensor, w: Tensor, num_nodes: int) -> Tensor:
    """out[i] = sum_{e: dst(e)=i} w_e * h[src(e)]   (message = w.view(-1,1) * x_j ; aggr = add)."""
    msg = w.view(-1, 1) * h.index_select(0, edge_index[0])          # materialises [E', F]
    return torch.zeros(num_nodes, h.size(1), dtype=h.dtype).index_add_(0, edge_index[1], msg)


def gcn_conv(
    x: Tensor,
    edge_index: Tensor,
    weight: Tensor,
    bias: Optional[Tensor],
    edge_weight: Optional[Tensor] = None,
    improved: bool = False,
    add_self_loops: bool = True,
    normalize: bool = True,
) -> Tensor:
    n = x.size(0)
    if normalize:
        ei, w = gcn_norm(edge_index, edge_weight, n, improved, add_self_loops, x.dtype)
    else:
        ei = edge_index
        w = edge_weight if edge_weight is not None else torch.ones(ei.size(1), dtype=x.dtype)
    h = x @ weight.t()                                              # lin: Linear(Fin,Fout,bias=False)
    out = propagate(h, ei, w.to(x.dtype), n)
    if bias is not None:
        out = out + bias
    return out


def glorot_(t: Tensor) -> Tensor:
    """PyG ``glorot``: U(-a, a), a = sqrt(6 / (fan_in + fan_out)) on a [out, in] matrix."""
    if t.numel() == 0:
        return t
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-a, a)


class _Lin(nn.Module):
    """Stand-in for PyG ``Linear(in, out, bias=False)``: exposes ``weight`` [out, in]."""

    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        glorot_(self.weight)


class OracleGCNConv(nn.Module):
    """Parameter layout of PyG GCNConv: ``bias`` then ``lin.weight``."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False,
                 cached: bool = False, add_self_loops: bool = True, normalize: bool = True,
                 bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.add_self_loops, self.normalize = improved, add_self_loops, normalize
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        self.lin = _Lin(in_channels, out_channels)

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
        return gcn_conv(x, edge_index, self.lin.weight, self.bias, edge_weight,
                        self.improved, self.add_self_loops, self.normalize)


# --------------------------------------------------------------------------------------------
# the model stack  (models_gnn.py:86-303)
# --------------------------------------------------------------------------------------------
@dataclass
class OracleGNNConfig:
    nodes_in: int
    nodes_out: int
    channels_in: int
    channels_out: int
    hidden_feats: int


class OracleDown(nn.Module):                      # models_gnn.py:106-157
    def __init__(self, c: OracleGNNConfig):
        super().__init__()
        h = c.hidden_feats
        self.conv1 = OracleGCNConv(c.channels_in, h)
        self.conv2 = OracleGCNConv(h, h // 2)
        self.conv3 = OracleGCNConv(h // 2, h // 4)
        self.conv4 = OracleGCNConv(h // 4, h // 8)      # declared, unused (:150)
        self.conv5 = OracleGCNConv(h // 8, h // 16)     # declared, unused (:151)

    def forward(self, x, ei):
        x = torch.relu(self.conv1(x, ei))
        x = torch.relu(self.conv2(x, ei))
        x = torch.relu(self.conv3(x, ei))
        return x


class OracleUp(nn.Module):                        # models_gnn.py:160-212
    def __init__(self, c: OracleGNNConfig):
        super().__init__()
        h = c.hidden_feats
        self.upconv1 = OracleGCNConv(h // 16, h // 8)   # declared, unused (:202)
        self.upconv2 = OracleGCNConv(h // 8, h // 4)    # declared, unused (:203)
        self.upconv3 = OracleGCNConv(h // 4, h // 2)
        self.upconv4 = OracleGCNConv(h // 2, h)
        self.upconv5 = OracleGCNConv(h, c.channels_out)

    def forward(self, x, ei):
        x = torch.relu(self.upconv3(x, ei))
        x = torch.relu(self.upconv4(x, ei))
        return self.upconv5(x, ei)


class OracleLayers(nn.Module):                    # models_gnn.py:215-258
    def __init__(self, c: OracleGNNConfig):
        super().__init__()
        self.down_conv_layers = OracleDown(c)
        self.up_conv_layers = OracleUp(c)

    def forward(self, x, ei):
        return self.up_conv_layers(self.down_conv_layers(x, ei), ei)


class OracleGNNModel(nn.Module):                  # models_gnn.py:268-303
    def __init__(self, c: OracleGNNConfig):
        super().__init__()
        self.conv_layers = OracleLayers(c)
        self.activation = nn.ReLU()

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        return self.conv_layers(x, edge_index)


def forward_with_masks(model: "OracleGNNModel", x: Tensor, edge_index: Tensor, masks) -> Tensor:
    """``OracleGNNModel.forward`` (models_gnn.py:135-157, :189-212) with every ReLU replaced by a GIVEN 0/1
    activation pattern: ``relu(pre)`` becomes ``pre * mask``.  Autograd through this is the gradient of the
    network AT that pattern -- the checker for a device backward that ran with the device's own patterns
    (a pre-activation within rounding of zero may fall on either side of the ReLU; with the pattern fixed, a
    flipped unit can no longer explain a difference).  ``masks``: five [N, F_l] tensors, one per ReLU."""
    d, u = model.conv_layers.down_conv_layers, model.conv_layers.up_conv_layers
    convs = [d.conv1, d.conv2, d.conv3, u.upconv3, u.upconv4]
    if len(masks) != len(convs):
        raise ValueError("one mask per ReLU (5)")
    for conv, m in zip(convs, masks):
        x = conv(x, edge_index) * m.to(x.dtype)
    return u.upconv5(x, edge_index)


def loss_func(output: Tensor, target: Tensor, target_mask: Tensor) -> Tensor:
    """L1 on masked rows -- the immediate consumer of forward (models_gnn.py:261-265)."""
    return nn.functional.l1_loss(output[target_mask], target[target_mask])


def processor_stack(x: Tensor, edge_index: Tensor, weights, biases) -> Tensor:
    """BASELINE config 3: chained F->F GCN layers with ReLU on one mesh (build-defined workload)."""
    for w, b in zip(weights, biases):
        x = torch.relu(gcn_conv(x, edge_index, w, b))
    return x
