"""CPU oracle of the BUILD-DEFINED grid -> mesh -> grid forecaster (gwen_amd/g2m.py)  --  TEST
INFRASTRUCTURE.  PARITY UNPINNED: the reference has no bipartite layers, no grid/mesh graphs and no
rollout, so there is nothing of the reference to follow or to pin against; this file
restates the semantics chosen in gwen_amd/g2m.py in plain torch so that the HIP path has a checker."""
from __future__ import annotations

import torch

from . import gcn_oracle as O


def bipartite_mean(x_src, edge_index, n_dst, weight, bias, relu=False):
    h = x_src @ weight.t()
    src, dst = edge_index[0], edge_index[1]
    deg = torch.zeros(n_dst, dtype=h.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=h.dtype))
    w = 1.0 / deg[dst]
    out = torch.zeros(n_dst, h.size(1), dtype=h.dtype).index_add_(0, dst, w.view(-1, 1) * h[src])
    if bias is not None:
        out = out + bias
    return torch.relu(out) if relu else out


def forward(state_dict, grid_x, g2m, mesh_ei, m2g, n_mesh, n_grid, steps):
    sd = state_dict
    h = bipartite_mean(grid_x, g2m, n_mesh, sd["encoder.lin.weight"], sd["encoder.bias"], relu=True)
    for k in range(steps):
        h = torch.relu(O.gcn_conv(h, mesh_ei, sd[f"processor.{k}.lin.weight"], sd[f"processor.{k}.bias"]))
    return bipartite_mean(h, m2g, n_grid, sd["decoder.lin.weight"], sd["decoder.bias"])
