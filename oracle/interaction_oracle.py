"""CPU oracle of the BUILD-DEFINED InteractionNet block (gwen_amd/interaction.py)  --  TEST
INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and benchmark checkers import it.

PARITY UNPINNED.  The reference has no edge MLP, no edge features and no encode-process-decode model:
its only graph layer is torch-geometric's GCNConv (/root/reference/src/gwen/models_gnn.py:118-130,
:147-149).  BASELINE.json's north_star names "the InteractionNet/GraphConv edge-MLP + scatter-add
node-aggregation block", so it is built with the published Interaction Network update (Battaglia et
al. 2016; the form used by encode-process-decode weather models) restated here in plain torch:

    m_e   = MLP_e([e, x_src[s(e)], x_dst[d(e)]])            MLP = Linear -> act -> Linear
    agg_d = sum (or mean) over edges e with d(e) = d of m_e    (edge order = stored order)
    x'_d  = x_dst_d + MLP_n([x_dst_d, agg_d])
    e'    = e + m_e

There is nothing of the reference to pin it against; the checker's own anchors are the closed-form
cases in tests/test_interaction.py (identity weights, isolated targets, linearity of the aggregate).
"""
from __future__ import annotations

import torch
from torch import Tensor


def act_fn(name: str):
    return {"none": lambda v: v, "relu": torch.relu, "silu": torch.nn.functional.silu}[name]


def mlp2(x: Tensor, w1: Tensor, b1, w2: Tensor, b2, act: str) -> Tensor:
    h = x @ w1.t()
    if b1 is not None:
        h = h + b1
    y = act_fn(act)(h) @ w2.t()
    return y if b2 is None else y + b2


def interaction(x_src: Tensor, x_dst: Tensor, e: Tensor, edge_index: Tensor, params: dict,
                act: str = "silu", aggr: str = "sum"):
    """``edge_index`` int64 [2,E] (row 0 = source, row 1 = target); ``e`` [E,F] in the same order.
    ``params``: edge_mlp.0.weight [F,3F], edge_mlp.0.bias, edge_mlp.2.weight [F,F], edge_mlp.2.bias,
    node_mlp.0.weight [F,2F], node_mlp.0.bias, node_mlp.2.weight, node_mlp.2.bias.
    Returns (x_dst', e')."""
    s, d = edge_index[0], edge_index[1]
    p = params
    m = mlp2(torch.cat([e, x_src[s], x_dst[d]], dim=1), p["edge_mlp.0.weight"], p["edge_mlp.0.bias"],
             p["edge_mlp.2.weight"], p["edge_mlp.2.bias"], act)
    agg = torch.zeros(x_dst.size(0), m.size(1), dtype=m.dtype).index_add_(0, d, m)
    if aggr == "mean":
        deg = torch.zeros(x_dst.size(0), dtype=m.dtype).index_add_(0, d, torch.ones(d.numel(), dtype=m.dtype))
        agg = agg / deg.clamp(min=1).view(-1, 1)
    x_new = x_dst + mlp2(torch.cat([x_dst, agg], dim=1), p["node_mlp.0.weight"], p["node_mlp.0.bias"],
                         p["node_mlp.2.weight"], p["node_mlp.2.bias"], act)
    return x_new, e + m


def _sub(sd: dict, prefix: str) -> dict:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def forecaster_step(sd: dict, grid_x: Tensor, mesh_pos: Tensor, g2m: Tensor, mesh_ei: Tensor, m2g: Tensor,
                    f_g2m: Tensor, f_mesh: Tensor, f_m2g: Tensor, steps: int, act: str = "silu",
                    aggr: str = "sum") -> Tensor:
    """One step of gwen_amd/forecaster.py's InteractionForecaster; edge lists and edge features in the
    SAME (caller's) order.  ``sd`` = its state_dict."""
    lin = lambda x, name: x @ sd[name + ".weight"].t() + sd[name + ".bias"]      # noqa: E731
    vg, vm = lin(grid_x, "grid_embed"), lin(mesh_pos, "mesh_embed")
    e_g2m, e_m, e_m2g = lin(f_g2m, "g2m_edge_embed"), lin(f_mesh, "mesh_edge_embed"), lin(f_m2g, "m2g_edge_embed")
    vm, _ = interaction(vg, vm, e_g2m, g2m, _sub(sd, "encoder."), act, aggr)
    for k in range(steps):
        vm, e_m = interaction(vm, vm, e_m, mesh_ei, _sub(sd, f"processor.{k}."), act, aggr)
    vg, _ = interaction(vm, vg, e_m2g, m2g, _sub(sd, "decoder."), act, aggr)
    return grid_x + lin(vg, "readout")
