"""Larger and skewed inputs on the device: beyond-L2 meshes, long rows, many members."""
import numpy as np
import pytest
import torch

from helpers import REL_TOL, SEED, make_params, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ga(hip_lib):
    import gwen_amd
    return gwen_amd


def test_nine_hundred_thousand_node_mesh_layer(ga):
    """nu = 300: N = 900 002, E = 5 400 000 (x = 230 MB at F = 64, beyond L2 and close to the Infinity
    Cache): one fused layer against the CPU oracle, linearity and determinism."""
    from oracle import gcn_oracle as O
    m = ga.geodesic_mesh(300)
    n = m.num_nodes
    assert (n, m.num_edges) == (900002, 5400000)
    ei = torch.from_numpy(m.edge_index)
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(64, 64)
    conv = ga.GCNConv(64, 64).to(DEV)
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
        eid = ei.to(DEV)
        got = conv(x.to(DEV), eid, relu=True)
        assert torch.equal(got, conv(x.to(DEV), eid, relu=True))
        ref = torch.relu(O.gcn_conv(x, ei, w, b))
    assert rel_err(got, ref) <= REL_TOL


def test_long_rows_complete_graph_1000(ga):
    """K_1000: 999 000 edges, every row 1000 entries (125 groups of 8) -- the long-row loops of K2/K4/K5."""
    n = 1000
    ei = torch.from_numpy(ga.complete_graph(n)).to(DEV)
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(64, 32)
    want = (x.double() @ w.double().t()).mean(0, keepdim=True) + b.double()       # K_N identity
    for order in ("auto", "fused_exact", "transform_first", "aggregate_first"):
        conv = ga.GCNConv(64, 32).to(DEV)
        conv.order = order
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.bias.copy_(b)
            got = conv(x.to(DEV), ei).cpu().double()
        assert rel_err(got, want.expand_as(got)) <= 2e-5, order


def test_star_graph_one_heavy_row(ga, cref):
    """One destination with 50 000 in-edges next to 50 000 single-entry rows (load skew)."""
    n = 50001
    ei = torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.long)])
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(32, 32)
    ref = torch.from_numpy(cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True))
    conv = ga.GCNConv(32, 32).to(DEV)
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
        got = conv(x.to(DEV), ei.to(DEV), relu=True).cpu()
    assert rel_err(got, ref) <= REL_TOL


def test_sixteen_members_share_one_graph(ga):
    m = ga.geodesic_mesh(20)
    ei = torch.from_numpy(m.edge_index).to(DEV)
    torch.manual_seed(SEED)
    model = ga.GNNModel(ga.GNNConfig(1, 1, 64, 64, 64)).to(DEV).eval()
    x = torch.randn(16, m.num_nodes, 64, device=DEV)
    with torch.no_grad():
        batched = model(x, ei)
        one = model(x[5], ei)
    assert torch.equal(batched[5], one)


def test_interaction_block_at_full_mesh_size(ga):
    """K6 at BASELINE's mesh size (nu = 100: 100 002 nodes, 600 000 edges) through size-independent
    properties: zero second layers leave the state untouched; the aggregate equals the target-wise sum
    of the edge updates (checked against torch's index_add_ on the device); two runs are bitwise equal."""
    from gwen_amd.interaction import InteractionNet, interaction_graph, mlp2
    F = 64
    m = ga.geodesic_mesh(100, reorder="morton")
    g = interaction_graph(torch.from_numpy(m.edge_index).to(DEV), m.num_nodes, m.num_nodes)
    assert g.num_edges == 600000 and g.max_degree == 6
    torch.manual_seed(SEED)
    net = InteractionNet(F).to(DEV)
    x = torch.randn(m.num_nodes, F, device=DEV)
    e = torch.randn(g.num_edges, F, device=DEV)
    with torch.no_grad():
        x1, e1 = net(x, x, e, g)
        x2, e2 = net(x, x, e, g)
        assert torch.equal(x1, x2) and torch.equal(e1, e2)
        w1 = net.edge_mlp[0].weight
        out, agg = mlp2(e, w1[:, :F].contiguous(), net.edge_mlp[2].weight, net.edge_mlp[2].bias, graph=g)
        want = torch.zeros(m.num_nodes, F, dtype=torch.float64, device=DEV).index_add_(0, g.dst.long(), out.double())
        assert float((agg.double() - want).abs().max() / want.abs().max()) <= 1e-6
        for p in (net.edge_mlp[2].weight, net.edge_mlp[2].bias, net.node_mlp[2].weight, net.node_mlp[2].bias):
            p.zero_()
        x3, e3 = net(x, x, e, g)
    assert torch.equal(x3, x) and torch.equal(e3, e)


@pytest.mark.parametrize("nu", [84, 100, 112])
@pytest.mark.parametrize("fin,fout", [(64, 64), (32, 64), (16, 32)])
def test_fused_layer_block_sizes_near_one_resident_round(ga, cref, nu, fin, fout):
    """K4 picks 64 / 96 / 112 / 128 rows per block so that a narrow layer runs as one round of co-resident
    blocks (N between ~65 000 and ~130 000 at 4 blocks per CU): every choice against the plain-C oracle
    (oracle/gcn_ref.c, fp64)."""
    from gwen_amd import ops
    m = ga.geodesic_mesh(nu)
    ei = torch.from_numpy(m.edge_index)
    g = ga.prepare_graph(ei.to(DEV), m.num_nodes)
    gen = torch.Generator().manual_seed(SEED + nu)
    x = torch.randn(m.num_nodes, fin, generator=gen)
    w = torch.rand(fout, fin, generator=gen) - 0.5
    b = torch.randn(fout, generator=gen)
    want = cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
    got = ops.layer_fused(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
    assert rel_err(got, want) <= 2e-5
    assert torch.equal(got, ops.layer_fused(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True))


def test_segment_chain_structure(ga):
    """gwen_gcn_segments: seg_rowptr refines rowptr in steps of 32 entries, rowptr2 counts each row's
    segments, the combine CSR is the identity pattern with unit weights; short-row graphs get no chain."""
    m = ga.geodesic_mesh(5)
    assert ga.prepare_graph(torch.from_numpy(m.edge_index).to(DEV), m.num_nodes).long_row_levels() is None
    assert ga.prepare_graph(torch.from_numpy(ga.complete_graph(150)).to(DEV), 150).long_row_levels() is None
    n = 50001
    ei = torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.long)]).to(DEV)      # one row of 50 001
    g = ga.prepare_graph(ei, n)
    lv = g.long_row_levels()
    assert lv is not None and len(lv) == 4            # 50 001 -> 1563 partials -> 49 -> 2 -> 1
    rowptr = g.rowptr.cpu().numpy().astype(np.int64)
    seg_rowptr, col, val, n_seg, cols = lv[0]
    sr = seg_rowptr.cpu().numpy().astype(np.int64)
    assert cols == n and n_seg == len(sr) - 1 and sr[0] == 0 and sr[-1] == rowptr[-1]
    assert set(rowptr.tolist()) <= set(sr.tolist()) and np.diff(sr).max() <= 32 and np.diff(sr).min() >= 1
    assert n_seg == (n - 1) + (50001 + 31) // 32
    rows_of_last = lv[-1][3]
    assert rows_of_last == n
    for (rp, c, v, rows, cols_), (_, _, _, rows_prev, _) in zip(lv[1:], lv[:-1]):
        assert cols_ == rows_prev
        k = int(rp[-1])
        assert torch.equal(c[:k].cpu(), torch.arange(k, dtype=torch.int32)) and bool((v[:k] == 1).all())


@pytest.mark.parametrize("case", ["K1000", "star", "hubs"])
def test_long_rows_edge_parallel_vs_oracle(ga, cref, case):
    """Rows far beyond 8 entries on graphs beyond K7's 256 nodes: the segment chain (32-entry segments summed
    by their own lane groups, partial sums added in segment order) against the C oracle; deterministic."""
    g = torch.Generator().manual_seed(SEED)
    if case == "K1000":
        n = 1000
        ei = torch.from_numpy(ga.complete_graph(n))
    elif case == "star":
        n = 50001
        ei = torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.long)])
    else:                                              # a mesh plus five hubs that every node points at
        m = ga.geodesic_mesh(20)
        n = m.num_nodes
        hubs = torch.randint(0, n, (5,), generator=g)
        extra = torch.stack([torch.arange(n).repeat(5), hubs.repeat_interleave(n)])
        ei = torch.cat([torch.from_numpy(m.edge_index), extra], 1)
    for fin, fout in ((64, 32), (32, 64), (20, 20)):
        x = torch.randn(n, fin, generator=g)
        w, b = make_params(fin, fout)
        ref = cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
        conv = ga.GCNConv(fin, fout).to(DEV)
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.bias.copy_(b)
            graph = ga.prepare_graph(ei.to(DEV), n)
            assert graph.long_row_levels() is not None
            got = conv(x.to(DEV), graph, relu=True)
            assert torch.equal(got, conv(x.to(DEV), graph, relu=True))
        assert rel_err(got, ref) <= 2e-5, (case, fin, fout)
    # the whole model (inference through StackForward's long-row path, training through the per-layer path)
    from oracle import gcn_oracle as O
    if case != "star":
        torch.manual_seed(SEED)
        refm = O.OracleGNNModel(O.OracleGNNConfig(n, n, 16, 16, 32))
        model = ga.GNNModel(ga.GNNConfig(n, n, 16, 16, 32))
        model.load_state_dict(refm.state_dict())
        model = model.to(DEV)
        x = torch.randn(n, 16, generator=g)
        with torch.no_grad():
            want = refm(x, ei)
            got = model(x.to(DEV), ei.to(DEV))
        assert rel_err(got, want) <= REL_TOL
        xd = x.to(DEV).requires_grad_()
        model(xd, ei.to(DEV)).sum().backward()
        xr = x.clone().requires_grad_()
        refm(xr, ei).sum().backward()
        assert float((xd.grad.cpu().double() - xr.grad.double()).norm() / xr.grad.double().norm()) <= 1e-3
