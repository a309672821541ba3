"""SURVEY 8(f) f2 (BUILD-DEFINED, parity unpinned): bipartite layers, grid->mesh->grid model, rollout."""
import numpy as np
import pytest
import torch

from helpers import REL_TOL, SEED, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ga(hip_lib):
    import gwen_amd
    return gwen_amd


def test_grid_mesh_edges_shape(ga):
    from gwen_amd import g2m
    m = ga.geodesic_mesh(5)
    a, b = g2m.grid_mesh_edges(m)
    assert a.shape == (2, 3 * 500) and np.array_equal(a[0], b[1]) and np.array_equal(a[1], b[0])
    assert np.bincount(a[0]).tolist() == [3] * 500            # every cell touches three vertices
    assert np.bincount(a[1], minlength=m.num_nodes).sum() == 1500


def test_rectangular_prep_is_mean_normalised(ga):
    from gwen_amd import g2m
    from gwen_amd.graph import prepare_bipartite
    m = ga.geodesic_mesh(4)
    a, _ = g2m.grid_mesh_edges(m)
    g = prepare_bipartite(torch.from_numpy(a).to(DEV), m.faces.shape[0], m.num_nodes)
    rp, val = g.rowptr.cpu().numpy(), g.val.cpu().numpy()
    assert rp[-1] == a.shape[1]
    for r in range(m.num_nodes):
        seg = val[rp[r]:rp[r + 1]]
        assert len(seg) in (5, 6) and abs(seg.sum() - 1) < 1e-6
    with pytest.raises(IndexError):
        prepare_bipartite(torch.tensor([[0, 9], [0, 1]], device=DEV), 5, 3)


@pytest.mark.parametrize("C,H,steps", [(16, 64, 2), (64, 64, 4), (24, 40, 1), (256, 256, 1)])
def test_model_and_rollout_vs_oracle(ga, C, H, steps):
    from gwen_amd import g2m
    from oracle import g2m_oracle as GO
    m = ga.geodesic_mesh(8)
    n_mesh, n_grid = m.num_nodes, m.faces.shape[0]
    torch.manual_seed(SEED)
    model = g2m.GridMeshGridModel(C, H, steps)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    graphs = model.prepare(m, DEV)
    x = torch.randn(n_grid, C, generator=torch.Generator().manual_seed(SEED))
    a, b = g2m.grid_mesh_edges(m)
    ei = torch.from_numpy(m.edge_index)
    with torch.no_grad():
        got = model(x.to(DEV), graphs).cpu()
    want = GO.forward(sd, x, torch.from_numpy(a), ei, torch.from_numpy(b), n_mesh, n_grid, steps)
    assert got.shape == (n_grid, C) and rel_err(got, want) <= REL_TOL
    states = model.rollout(x.to(DEV), graphs, 3)
    cur = x
    for s in states:
        cur = GO.forward(sd, cur, torch.from_numpy(a), ei, torch.from_numpy(b), n_mesh, n_grid, steps)
        assert rel_err(s, cur) <= 3 * REL_TOL


@pytest.mark.parametrize("C,H", [(16, 64), (24, 40)])
def test_grid_mesh_grid_model_trains(ga, C, H):
    """Gradients of the grid -> mesh -> grid model (bipartite encoder / decoder on the rectangular transposes,
    GCN processor) against torch autograd on the CPU oracle; and one Adam step lowers the loss."""
    from gwen_amd import g2m
    from oracle import g2m_oracle as GO
    m = ga.geodesic_mesh(6, reorder="hilbert")
    n_mesh, n_grid, steps = m.num_nodes, m.faces.shape[0], 2
    torch.manual_seed(SEED)
    model = g2m.GridMeshGridModel(C, H, steps)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    sd = {k: v.clone().double().requires_grad_() for k, v in model.state_dict().items()}
    a, b = g2m.grid_mesh_edges(m)
    x = torch.randn(n_grid, C, generator=torch.Generator().manual_seed(SEED))
    gout = torch.randn(n_grid, C, generator=torch.Generator().manual_seed(SEED + 1))
    xr = x.double().requires_grad_()
    GO.forward(sd, xr, torch.from_numpy(a), torch.from_numpy(m.edge_index), torch.from_numpy(b), n_mesh, n_grid,
               steps).backward(gout.double())
    model = model.to(DEV)
    graphs = model.prepare(m, DEV)
    xd = x.to(DEV).requires_grad_()
    out = model(xd, graphs)
    out.backward(gout.to(DEV))

    def l2(a_, b_):
        return float((a_.double().cpu() - b_).norm() / b_.norm())
    assert l2(xd.grad, xr.grad) <= 1e-3                    # ReLU flips allowed for (see test_gpu_parity.py)
    for k, p in model.named_parameters():
        assert l2(p.grad, sd[k].grad) <= 1e-3, k
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = (model(x.to(DEV), graphs) - gout.to(DEV)).abs().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]


def test_rectangular_transpose(ga):
    from gwen_amd import g2m
    from gwen_amd.graph import prepare_bipartite
    m = ga.geodesic_mesh(4)
    a, _ = g2m.grid_mesh_edges(m)
    n_grid, n_mesh = m.faces.shape[0], m.num_nodes
    g = prepare_bipartite(torch.from_numpy(a).to(DEV), n_grid, n_mesh)
    t = g.transposed_graph()
    assert (t.num_nodes, t.source_nodes) == (n_grid, n_mesh)
    dense = torch.zeros(n_mesh, n_grid, dtype=torch.float64)
    rp, col, val = g.rowptr.cpu(), g.col.cpu(), g.val.cpu()
    for r in range(n_mesh):
        for s_ in range(int(rp[r]), int(rp[r + 1])):
            dense[r, int(col[s_])] += float(val[s_])
    dt = torch.zeros(n_grid, n_mesh, dtype=torch.float64)
    rp, col, val = t.rowptr.cpu(), t.col.cpu(), t.val.cpu()
    for r in range(n_grid):
        for s_ in range(int(rp[r]), int(rp[r + 1])):
            dt[r, int(col[s_])] += float(val[s_])
    assert torch.equal(dt, dense.t())
