"""Shared test inputs and error measures (CPU side)."""
from __future__ import annotations

import numpy as np
import torch

SEED = 23   # the reference's seed: /root/reference/src/gwen/config.json:14
REL_TOL = 1e-4   # BASELINE.json north_star: "within 1e-4 rel-fp32"


def rel_err(a, b) -> float:
    """max |a - b| / max |b|  (relative to the tensor's scale, as for an fp32 matmul chain)."""
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64) if not isinstance(a, torch.Tensor) else a.double().cpu()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64) if not isinstance(b, torch.Tensor) else b.double().cpu()
    if b.numel() == 0:
        return 0.0
    scale = float(b.abs().max())
    return float((a - b).abs().max()) / (scale if scale > 0 else 1.0)


def random_multigraph(n: int, e: int, seed: int = SEED, self_loops: int = 0, dup: int = 0,
                      isolate: int = 0) -> torch.Tensor:
    """Random directed multigraph [2, E] with optional explicit self-loops, duplicated edges and
    `isolate` nodes (the highest ids) that receive no edge at all."""
    g = torch.Generator().manual_seed(seed)
    hi = max(n - isolate, 1)
    ei = torch.randint(0, hi, (2, e), generator=g)
    parts = [ei]
    if self_loops:
        s = torch.randint(0, hi, (self_loops,), generator=g)
        parts.append(torch.stack([s, s]))
    if dup and e:
        idx = torch.randint(0, e, (dup,), generator=g)
        parts.append(ei[:, idx])
    ei = torch.cat(parts, 1)
    perm = torch.randperm(ei.size(1), generator=g)
    return ei[:, perm].contiguous()


def graph_cases():
    """(name, num_nodes, edge_index) covering the edge cases the domain has."""
    from gwen_amd.mesh import complete_graph, geodesic_mesh
    cases = []
    m = geodesic_mesh(10)
    cases.append(("mesh_c1", m.num_nodes, torch.from_numpy(m.edge_index)))
    cases.append(("K125", 125, torch.from_numpy(complete_graph(125))))        # reference's family
    cases.append(("K7", 7, torch.from_numpy(complete_graph(7))))
    cases.append(("empty", 5, torch.zeros(2, 0, dtype=torch.long)))
    cases.append(("one_node", 1, torch.zeros(2, 0, dtype=torch.long)))
    cases.append(("path3", 3, torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])))
    cases.append(("cycle6", 6, torch.tensor([[0, 1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 0]])))
    cases.append(("multi", 300, random_multigraph(300, 2000, self_loops=40, dup=100, isolate=7)))
    cases.append(("only_loops", 4, torch.tensor([[0, 1, 1, 3], [0, 1, 1, 3]])))
    cases.append(("star", 200, torch.stack([torch.arange(1, 200), torch.zeros(199, dtype=torch.long)])))
    return cases


def make_params(fin: int, fout: int, seed: int = SEED):
    g = torch.Generator().manual_seed(seed + 1000 * fin + fout)
    a = (6.0 / (fin + fout)) ** 0.5 if fin + fout > 0 else 0.0
    w = (torch.rand(fout, fin, generator=g) * 2 - 1) * a
    b = torch.randn(fout, generator=g) * 0.1
    return w.contiguous(), b.contiguous()


def csr_from_oracle(src, dst, w, n):
    """Oracle edge list (PyG order) -> per-row lists in visiting order (stable by target)."""
    order = np.argsort(dst, kind="stable")
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, src[order], w[order]
