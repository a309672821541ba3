"""Encode-process-decode forecaster on InteractionNet blocks (BASELINE config c5).

BUILD-DEFINED, PARITY UNPINNED -- the reference has neither grid/mesh graphs nor edge MLPs nor a
rollout (its model is six GCNConv calls, /root/reference/src/gwen/models_gnn.py:
135-157,:189-212).  BASELINE.json's north_star names "the InteractionNet/GraphConv edge-MLP +
scatter-add node-aggregation block that propagates atmospheric state over the grid->mesh->grid graphs
each rollout step"; this module is that loop, with semantics of this build's own choosing (restated on
the CPU in oracle/interaction_oracle.py):

    vg  = grid_x  Wg^T + bg                 vm = mesh_pos Wm^T + bm              (embedders, K3)
    e_* = edge_feat_* We_*^T + be_*         edge_feat = [length, dx, dy, dz] of the edge
    vm      = Encoder(vg, vm, e_g2m)        InteractionNet, grid -> mesh   (gwen_amd/interaction.py, K6)
    vm, e_m = Processor_k(vm, vm, e_m)      InteractionNet, mesh -> mesh,  k = 1..steps
    vg      = Decoder(vm, vg, e_m2g)        InteractionNet, mesh -> grid
    grid_y  = grid_x + vg Wo^T + bo                                          (residual read-out, K3)

The grid is the set of triangle centres of the geodesic mesh; every cell is linked with its three
corner vertices (gwen_amd/g2m.py).  Static embeddings (vm, e_*) depend on the weights only and are
computed once per ``forward`` / ``rollout`` call.  Trainable (``interaction._InteractionNetFunction``).  A leading members axis
``[members, N_grid, C]`` runs as ONE launch set over the block-diagonal graph (``ForecastGraphs.batched``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import ops
from .g2m import grid_mesh_edges
from .interaction import EdgeGraph, InteractionNet, interaction_graph
from .mesh import Mesh


def edge_features(pos_src: np.ndarray, pos_dst: np.ndarray, edge_index: np.ndarray) -> np.ndarray:
    """[E, 4] float32: chord length and displacement (target - source) of every edge."""
    d = pos_dst[edge_index[1]] - pos_src[edge_index[0]]
    return np.concatenate([np.linalg.norm(d, axis=1, keepdims=True), d], axis=1).astype(np.float32)


@dataclass
class ForecastGraphs:
    g2m: EdgeGraph
    mesh: EdgeGraph
    m2g: EdgeGraph
    mesh_pos: Tensor        # [Nm, 3]
    f_g2m: Tensor           # [E, 4] each, in the STORED edge order of its graph
    f_mesh: Tensor
    f_m2g: Tensor
    _batched: dict = None   # members -> ForecastGraphs of the block-diagonal graphs

    def batched(self, members: int) -> "ForecastGraphs":
        """Graphs and static inputs of ``members`` independent copies (EdgeGraph.batched): the c5 path rolls
        every local member through ONE launch set per step."""
        if members == 1:
            return self
        if self._batched is None:
            self._batched = {}
        if members not in self._batched:
            rep = lambda t: t.repeat(members, 1)                                   # noqa: E731
            self._batched[members] = ForecastGraphs(
                self.g2m.batched(members), self.mesh.batched(members), self.m2g.batched(members),
                rep(self.mesh_pos), rep(self.f_g2m), rep(self.f_mesh), rep(self.f_m2g))
        return self._batched[members]


class InteractionForecaster(nn.Module):
    def __init__(self, grid_channels: int, hidden: int, steps: int = 4, activation: str = "silu",
                 aggr: str = "sum"):
        super().__init__()
        self.grid_channels, self.hidden, self.steps = grid_channels, hidden, steps
        self.grid_embed = nn.Linear(grid_channels, hidden)
        self.mesh_embed = nn.Linear(3, hidden)
        self.g2m_edge_embed = nn.Linear(4, hidden)
        self.mesh_edge_embed = nn.Linear(4, hidden)
        self.m2g_edge_embed = nn.Linear(4, hidden)
        self.encoder = InteractionNet(hidden, activation, aggr)
        self.processor = nn.ModuleList([InteractionNet(hidden, activation, aggr) for _ in range(steps)])
        self.decoder = InteractionNet(hidden, activation, aggr)
        self.readout = nn.Linear(hidden, grid_channels)

    @staticmethod
    def prepare(mesh: Mesh, device) -> ForecastGraphs:
        g2m, m2g = grid_mesh_edges(mesh)
        n_mesh, n_grid = mesh.num_nodes, mesh.faces.shape[0]
        cell = mesh.pos[mesh.faces].mean(axis=1)
        cell /= np.linalg.norm(cell, axis=1, keepdims=True)
        gs = (interaction_graph(torch.from_numpy(g2m).to(device), n_grid, n_mesh),
              interaction_graph(torch.from_numpy(mesh.edge_index).to(device), n_mesh, n_mesh),
              interaction_graph(torch.from_numpy(m2g).to(device), n_mesh, n_grid))
        feats = (edge_features(cell, mesh.pos, g2m), edge_features(mesh.pos, mesh.pos, mesh.edge_index),
                 edge_features(mesh.pos, cell, m2g))
        fs = [g.sort_edges(torch.from_numpy(f).to(device)) for g, f in zip(gs, feats)]
        return ForecastGraphs(*gs, torch.from_numpy(mesh.pos.astype(np.float32)).to(device), *fs)

    @staticmethod
    def _lin(x: Tensor, m: nn.Linear) -> Tensor:
        """K3 (bf16x3, as the blocks around it); with autograd when gradients are needed (ops.LinearFunction:
        the backward runs on K3 and the gradient reductions too)."""
        if torch.is_grad_enabled() and (x.requires_grad or m.weight.requires_grad):
            return ops.linear_autograd(x, m.weight, m.bias, contract="3xbf16")
        return ops.linear(x, m.weight, m.bias, exact=False)

    def _static(self, graphs: ForecastGraphs):
        lin = self._lin
        return (lin(graphs.mesh_pos, self.mesh_embed), lin(graphs.f_g2m, self.g2m_edge_embed),
                lin(graphs.f_mesh, self.mesh_edge_embed), lin(graphs.f_m2g, self.m2g_edge_embed))

    def _step(self, grid_x: Tensor, graphs: ForecastGraphs, static, out: Optional[Tensor] = None) -> Tensor:
        vm, e_g2m, e_m, e_m2g = static
        vg = self._lin(grid_x, self.grid_embed)
        vm, _ = self.encoder(vg, vm, e_g2m, graphs.g2m, update_edges=False)
        for net in self.processor:
            vm, e_m = net(vm, vm, e_m, graphs.mesh)
        vg, _ = self.decoder(vm, vg, e_m2g, graphs.m2g, update_edges=False)
        delta = self._lin(vg, self.readout)
        return grid_x + delta if out is None else torch.add(grid_x, delta, out=out)      # (out: GraphedStep's buffers)

    def forward(self, grid_x: Tensor, graphs: ForecastGraphs) -> Tensor:
        """``grid_x`` [N_grid, C] or [members, N_grid, C] (members share graphs and weights: one launch set
        over the block-diagonal graph)."""
        if grid_x.dim() == 3:
            m = grid_x.size(0)
            gb = graphs.batched(m)
            return self._step(grid_x.reshape(-1, grid_x.size(-1)), gb, self._static(gb)).view_as(grid_x)
        return self._step(grid_x, graphs, self._static(graphs))

    def rollout(self, grid_x: Tensor, graphs: ForecastGraphs, n_steps: int,
                graphed: bool = False) -> List[Tensor]:
        """Autoregressive: state_{t+1} = forward(state_t); returns the n_steps states.
        ``graphed``: capture ONE step (its ~26 launches) into a hipGraph and replay it per step -- the
        launchers allocate and synchronise nothing, so the step is capturable as is; worth it when the
        host cannot keep ahead of the device (64 channels: 1.31 -> 1.14 ms per step; 128: 2.71 -> 2.39)."""
        states, cur = [], grid_x
        with torch.no_grad():
            if graphed:
                step = GraphedStep(self, graphs, grid_x)
                for _ in range(n_steps):
                    cur = step(cur)                          # (one of the step's two buffers: the next call reads it in place)
                    states.append(cur.clone())
                return states
            static = self._static(graphs)
            for _ in range(n_steps):
                cur = self._step(cur, graphs, static)
                states.append(cur)
        return states


def ensemble_forecast(model, graphs: ForecastGraphs, x_members: Tensor,
                      n_steps: int, num_members: int, group=None, graphed: bool = True,
                      batched: bool = True, step_cache: dict = None, gather: bool = True) -> Tensor:
    """BASELINE config c5: this rank's members ``[members_local, N_grid, C]`` are rolled out ``n_steps``
    steps (independent members, replicated graph and weights), then every rank's final states are gathered
    ONCE (``ensemble.gather_members``: RCCL all-gather over xGMI under the "nccl" backend).  Returns
    ``[num_members, N_grid, C]`` on every rank.

    ``batched`` (default): all local members advance together -- ONE launch set per step over the
    block-diagonal graph (``ForecastGraphs.batched``), captured once and replayed when ``graphed``.
    ``batched=False``: one member after the other (the same arithmetic; kept for comparison).
    ``step_cache``: a dict the captured step lives in between calls (capture costs one eager step plus the
    capture itself; the static embeddings inside it are those of the weights at capture time).
    ``gather=False`` returns this rank's final states ``[members_local, N_grid, C]`` without the collective.
    ``model`` needs ``_static(graphs)`` and ``_step(x, graphs, static)`` (InteractionForecaster)."""
    from . import ensemble
    m_local = x_members.size(0)

    def captured(g, x0):
        if step_cache is None:
            return GraphedStep(model, g, x0)
        key = (id(model), id(g), tuple(x0.shape))
        if key not in step_cache:
            step_cache[key] = GraphedStep(model, g, x0)
        return step_cache[key]

    with torch.no_grad():
        if m_local == 0:
            local = x_members.new_empty((0,) + tuple(x_members.shape[1:]))
        elif batched:
            gb = graphs.batched(m_local)
            cur = x_members.reshape(-1, x_members.size(-1))
            if graphed:
                step = captured(gb, cur)
                for _ in range(n_steps):             # (the step alternates between its two state buffers: no copies)
                    cur = step(cur)
                cur = cur.clone()
            else:
                static = model._static(gb)
                for _ in range(n_steps):
                    cur = model._step(cur, gb, static)
            local = cur.view_as(x_members)
        else:
            finals = []
            step = captured(graphs, x_members[0]) if graphed else None
            static = None if step is not None else model._static(graphs)
            for m in range(m_local):
                cur = x_members[m]
                for _ in range(n_steps):
                    cur = step(cur) if step is not None else model._step(cur, graphs, static)
                finals.append(cur.clone() if step is not None else cur)
            local = torch.stack(finals)
    if not gather:
        return local
    return ensemble.gather_members(local, num_members, group)


class GraphedStep:
    """One forecaster step captured into hipGraphs (``torch.cuda.CUDAGraph``) over TWO state buffers -- the graph A -> B
    and the graph B -> A -- so that an autoregressive rollout replays them alternately without copying the state (one
    capture over a fixed input / output pair cost two state-sized copies per step: 2 % of the c5 rollout).  ``step(x)``
    copies ``x`` into the next input buffer unless it IS that buffer (the previous call's result), replays, and returns the
    output buffer -- valid until the call AFTER the next one (clone it to keep it longer)."""

    def __init__(self, model: InteractionForecaster, graphs: ForecastGraphs, grid_x: Tensor):
        self.bufs = [grid_x.detach().clone(), torch.empty_like(grid_x)]
        self.graphs = graphs                                 # the captured graphs hold raw pointers into these:
        self.cur = 0                                         # the buffer the next call reads
        with torch.no_grad():                                # keep every tensor they read alive
            static = self._static = model._static(graphs)
            model._step(self.bufs[0], graphs, static, out=self.bufs[1])       # warm-up: occupancy queries, tilings, caches
            torch.cuda.synchronize(grid_x.device)
            self.replays = []
            for i in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    model._step(self.bufs[i], graphs, static, out=self.bufs[1 - i])
                self.replays.append(g)

    def __call__(self, x: Tensor) -> Tensor:
        src = self.bufs[self.cur]
        if x.data_ptr() != src.data_ptr():
            src.copy_(x)
        self.replays[self.cur].replay()
        self.cur ^= 1
        return self.bufs[self.cur]
