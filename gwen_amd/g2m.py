"""Grid -> mesh -> grid forecaster on the same kernels (BASELINE config c5).

BUILD-DEFINED.  The reference has no grid/mesh graphs, no bipartite layers and no autoregressive
rollout (its graph is the complete graph over ensemble members and its loop is one
forward per batch); BASELINE.json names them, so they are built here with semantics of this build's own
choosing and an oracle of this build's own writing (oracle/g2m_oracle.py) -- PARITY UNPINNED, nothing in
the reference to compare with:

    encoder   mesh_h  = ReLU( mean_{g in N(m)} (grid_x[g] We^T) + be )         grid -> mesh, bipartite
    processor mesh_h  = ReLU( GCNConv_k(mesh_h) )  for k = 1..steps           mesh -> mesh (K4 layers)
    decoder   grid_y  = mean_{m in N(g)} (mesh_h[m] Wd^T) + bd                mesh -> grid, bipartite

The grid is the set of triangle centres of the geodesic mesh (20 nu^2 cells); every cell is linked to its
three corner vertices in both directions.  A bipartite layer is one K4 launch (or K3 + K2)
over a rectangular CSR (gwen_gcn_prep_rect); the rollout feeds each step's grid output back as input.
Trainable: the bipartite layers go through the GCN layer's autograd Function.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import Tensor, nn

from . import ops
from .gcn_conv import GCNConv, Linear
from .graph import GraphCSR, prepare_bipartite, prepare_graph
from .mesh import Mesh


def grid_mesh_edges(mesh: Mesh):
    """(g2m, m2g) edge lists int64 [2, 3 * faces]: grid cell f <-> its three mesh vertices."""
    f = mesh.faces
    cells = np.repeat(np.arange(f.shape[0], dtype=np.int64), 3)
    verts = f.reshape(-1).astype(np.int64)
    g2m = np.stack([cells, verts])          # source = grid cell, target = mesh vertex
    m2g = np.stack([verts, cells])          # source = mesh vertex, target = grid cell
    return g2m, m2g


class BipartiteConv(nn.Module):
    """``out_dst = act( mean over in-edges of (x_src W^T) + b )`` -- parameters ``lin.weight``, ``bias``."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = Linear(in_channels, out_channels)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def forward(self, x_src: Tensor, graph: GraphCSR, relu: bool = False) -> Tensor:
        """The GCN layer's kernels on a rectangular graph, differentiable through the same autograd Function
        (its backward walks the rectangular transpose: ``GraphCSR.transposed_graph``)."""
        return ops.gcn_layer(x_src, self.lin.weight, self.bias, graph, relu=relu, order="auto")


class GridMeshGridModel(nn.Module):
    def __init__(self, grid_channels: int, hidden: int, steps: int = 4):
        super().__init__()
        self.encoder = BipartiteConv(grid_channels, hidden)
        self.processor = nn.ModuleList([GCNConv(hidden, hidden) for _ in range(steps)])
        self.decoder = BipartiteConv(hidden, grid_channels)

    def prepare(self, mesh: Mesh, device) -> "GridMeshGraphs":
        g2m, m2g = grid_mesh_edges(mesh)
        n_mesh, n_grid = mesh.num_nodes, mesh.faces.shape[0]
        return GridMeshGraphs(
            g2m=prepare_bipartite(torch.from_numpy(g2m).to(device), n_grid, n_mesh),
            mesh=prepare_graph(torch.from_numpy(mesh.edge_index).to(device), n_mesh),
            m2g=prepare_bipartite(torch.from_numpy(m2g).to(device), n_mesh, n_grid))

    def forward(self, grid_x: Tensor, graphs: "GridMeshGraphs") -> Tensor:
        h = self.encoder(grid_x, graphs.g2m, relu=True)
        for conv in self.processor:
            h = conv(h, graphs.mesh, relu=True)
        return self.decoder(h, graphs.m2g)

    def rollout(self, grid_x: Tensor, graphs: "GridMeshGraphs", n_steps: int) -> List[Tensor]:
        """Autoregressive: state_{t+1} = forward(state_t); returns the n_steps states."""
        states, cur = [], grid_x
        with torch.no_grad():
            for _ in range(n_steps):
                cur = self.forward(cur, graphs)
                states.append(cur)
        return states


class GridMeshGraphs:
    def __init__(self, g2m: GraphCSR, mesh: GraphCSR, m2g: GraphCSR):
        self.g2m, self.mesh, self.m2g = g2m, mesh, m2g
