"""SURVEY 8(f) f2 (BUILD-DEFINED, parity unpinned): K6 -- edge MLP + sum to targets + node MLP
(gwen_mlp2_f32, gwen_edge_tiles) against oracle/interaction_oracle.py; fp32 tolerance 1e-4 relative,
index structures bit-exact."""
import numpy as np
import pytest
import torch

from helpers import REL_TOL, SEED, random_multigraph, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ga(hip_lib):
    import gwen_amd
    return gwen_amd


def _graphs(ga):
    from gwen_amd import g2m
    from gwen_amd.mesh import complete_graph
    m = ga.geodesic_mesh(6)
    a, b = g2m.grid_mesh_edges(m)
    n, nf = m.num_nodes, m.faces.shape[0]
    return {
        "mesh": (n, n, torch.from_numpy(m.edge_index)),
        "g2m": (nf, n, torch.from_numpy(a)),
        "m2g": (n, nf, torch.from_numpy(b)),
        "K125": (125, 125, torch.from_numpy(complete_graph(125))),       # rows longer than one pass
        "multi": (300, 300, random_multigraph(300, 2000, self_loops=40, dup=100, isolate=7)),
        "sparse": (50, 4000, random_multigraph(50, 300, seed=5)[:, :300] * torch.tensor([[1], [80]])),
        "star": (200, 200, torch.stack([torch.arange(1, 200), torch.zeros(199, dtype=torch.long)])),
        "empty": (5, 7, torch.zeros(2, 0, dtype=torch.long)),
        "one_edge": (3, 3, torch.tensor([[2], [1]])),
    }


@pytest.mark.parametrize("rows", [64, 32])
@pytest.mark.parametrize("name", ["mesh", "g2m", "m2g", "K125", "multi", "sparse", "star", "empty", "one_edge"])
def test_edge_tiles_are_row_aligned(ga, name, rows):
    from gwen_amd.interaction import interaction_graph
    ns, nd, ei = _graphs(ga)[name]
    g = interaction_graph(ei.to(DEV), ns, nd)
    rp = g.rowptr.cpu().numpy().astype(np.int64)
    e = ei.size(1)
    assert rp[0] == 0 and rp[-1] == e and g.num_edges == e
    # stored order = stable sort by target; src/dst/eid describe the same edges
    order = np.argsort(ei[1].numpy(), kind="stable")
    assert np.array_equal(g.dst.cpu().numpy(), ei[1].numpy()[order])
    eid = g.eid.cpu().numpy()
    assert np.array_equal(np.sort(eid), np.arange(e))
    assert np.array_equal(g.src.cpu().numpy(), ei[0].numpy()[eid])
    assert np.array_equal(g.dst.cpu().numpy(), ei[1].numpy()[eid])
    # tiles: contiguous row ranges covering every row once; tile c starts at the first row whose
    # first edge is at or after c T
    tile_row, n_tiles = g.tiles(rows)
    tr = tile_row.cpu().numpy()
    assert tr[0] == 0 and tr[-1] == nd and np.all(np.diff(tr) >= 0) and len(tr) == n_tiles + 1
    max_deg = int(np.diff(rp).max()) if nd else 0
    assert g.max_degree == max_deg
    spans = rp[tr[1:]] - rp[tr[:-1]]
    if max_deg <= rows // 2 + 1:
        assert spans.max() <= rows                       # a tile is a single pass of the kernel
    t = max(rows - max(max_deg - 1, 0), rows // 2) if max_deg <= rows else rows
    firsts = rp[tr[:-1]]                                 # tile c starts at the first row at or after c T
    assert all(firsts[c] >= c * t and (tr[c] == 0 or rp[tr[c] - 1] < c * t) for c in range(1, n_tiles))
    x = torch.randn(e, 3)
    assert torch.equal(g.unsort_edges(g.sort_edges(x.to(DEV))).cpu(), x)


@pytest.mark.parametrize("F", [32, 64, 128, 256])
@pytest.mark.parametrize("rows", [1, 63, 64, 65, 1000])
@pytest.mark.parametrize("act", ["none", "relu", "silu"])
def test_row_mlp_vs_oracle(ga, F, rows, act):
    from gwen_amd.interaction import mlp2
    from oracle import interaction_oracle as IO
    g = torch.Generator().manual_seed(SEED + F + rows)
    a = torch.randn(rows, F, generator=g)
    w1 = torch.randn(F, F, generator=g) / F ** 0.5
    w2 = torch.randn(F, F, generator=g) / F ** 0.5
    b1, b2 = torch.randn(F, generator=g), torch.randn(F, generator=g)
    tab = torch.randn(17, F, generator=g)
    idx = torch.randint(0, 17, (rows,), generator=g)
    want = a + IO.act_fn(act)(a.double() @ w1.double().t() + tab[idx].double() + b1.double()) @ w2.double().t() + b2
    got, agg = mlp2(a.to(DEV), w1.to(DEV), w2.to(DEV), b2.to(DEV), g1=tab.to(DEV),
                    idx1=idx.to(torch.int32).to(DEV), b1=b1.to(DEV), res=a.to(DEV), act=act)
    assert agg is None
    assert rel_err(got, want) <= REL_TOL


def _params(F, seed):
    from gwen_amd.interaction import InteractionNet
    torch.manual_seed(seed)
    net = InteractionNet(F)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    return net


@pytest.mark.parametrize("name", ["mesh", "g2m", "m2g", "K125", "multi", "sparse", "star", "empty", "one_edge"])
@pytest.mark.parametrize("F,act,aggr", [(64, "silu", "sum"), (32, "relu", "mean"), (128, "silu", "sum"), (256, "silu", "mean")])
def test_interaction_block_vs_oracle(ga, name, F, act, aggr):
    from gwen_amd.interaction import InteractionNet, interaction_graph
    from oracle import interaction_oracle as IO
    ns, nd, ei = _graphs(ga)[name]
    torch.manual_seed(SEED)
    net = InteractionNet(F, act, aggr)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    g = torch.Generator().manual_seed(SEED + 1)
    xs = torch.randn(ns, F, generator=g)
    xd = xs if name in ("mesh", "K125", "multi", "star", "one_edge") else torch.randn(nd, F, generator=g)
    e = torch.randn(ei.size(1), F, generator=g)
    sd = {k: v.double() for k, v in net.state_dict().items()}
    want_x, want_e = IO.interaction(xs.double(), xd.double(), e.double(), ei, sd, act, aggr)
    graph = interaction_graph(ei.to(DEV), ns, nd)
    net = net.to(DEV)
    xs_d = xs.to(DEV)
    xd_d = xs_d if xd is xs else xd.to(DEV)      # same object: one stacked projection launch
    with torch.no_grad():
        got_x, got_e = net(xs_d, xd_d, graph.sort_edges(e.to(DEV)), graph)
        again_x, again_e = net(xs_d, xd_d, graph.sort_edges(e.to(DEV)), graph)
        apart_x, apart_e = net(xs_d, xd_d.clone(), graph.sort_edges(e.to(DEV)), graph)   # sources apart
    assert rel_err(got_x, want_x) <= REL_TOL
    assert rel_err(graph.unsort_edges(got_e), want_e) <= REL_TOL
    assert torch.equal(got_x, again_x) and torch.equal(got_e, again_e)        # no atomics anywhere
    assert rel_err(apart_x, want_x) <= REL_TOL and rel_err(graph.unsort_edges(apart_e), want_e) <= REL_TOL


def test_closed_forms(ga):
    """Anchors that need no oracle: zero second layers leave the state untouched; targets without
    in-edges receive an empty sum; the aggregate is the sum of the edge updates."""
    from gwen_amd.interaction import InteractionNet, interaction_graph, mlp2
    F = 64
    ns, nd, ei = _graphs(ga)["sparse"]
    graph = interaction_graph(ei.to(DEV), ns, nd)
    net = _params(F, 3).to(DEV)
    xs, xd = torch.randn(ns, F, device=DEV), torch.randn(nd, F, device=DEV)
    e = torch.randn(graph.num_edges, F, device=DEV)
    with torch.no_grad():
        net.edge_mlp[2].weight.zero_(); net.edge_mlp[2].bias.zero_()
        net.node_mlp[2].weight.zero_(); net.node_mlp[2].bias.zero_()
        x1, e1 = net(xs, xd, e, graph)
    assert torch.equal(x1, xd) and torch.equal(e1, e)
    # agg == segmented sum of (e' - e), and rows without in-edges are exactly zero
    net = _params(F, 4).to(DEV)
    with torch.no_grad():
        w1 = net.edge_mlp[0].weight
        out, agg = mlp2(e, w1[:, :F], net.edge_mlp[2].weight, net.edge_mlp[2].bias, act="silu",
                        graph=graph)
    m = out.double().cpu()
    want = torch.zeros(nd, F, dtype=torch.float64).index_add_(0, graph.dst.long().cpu(), m)
    assert rel_err(agg, want) <= 1e-6
    deg = np.diff(graph.rowptr.cpu().numpy())
    assert (deg == 0).sum() > 0
    assert torch.count_nonzero(agg[torch.from_numpy(deg == 0).to(DEV)]) == 0


def test_rejects_misuse(ga):
    from gwen_amd.interaction import InteractionNet, interaction_graph, mlp2
    a = torch.randn(10, 48, device=DEV)
    with pytest.raises(ValueError):
        mlp2(a, torch.randn(48, 48, device=DEV), torch.randn(48, 48, device=DEV))
    a = torch.randn(10, 64, device=DEV)
    w = torch.randn(64, 64, device=DEV)
    wide = torch.randn(10, 192, device=DEV)
    out_a, _ = mlp2(a, w, w, g1=wide[:, 64:128])                 # a column block of a wider matrix
    out_b, _ = mlp2(a, w, w, g1=wide[:, 64:128].contiguous())
    assert torch.equal(out_a, out_b)
    with pytest.raises(ValueError):
        mlp2(a, w, w, g1=wide[:, 2:66])                          # rows not 16-byte aligned
    with pytest.raises(ValueError):
        mlp2(a, w, w, g1=torch.randn(5, 64, device=DEV))
    with pytest.raises(ValueError):
        mlp2(a, w, w, g1=torch.randn(5, 64, device=DEV), idx1=torch.zeros(10, dtype=torch.int64, device=DEV))
    ns, nd, ei = _graphs(ga)["one_edge"]
    graph = interaction_graph(ei.to(DEV), ns, nd)
    net = InteractionNet(64).to(DEV)
    with pytest.raises(ValueError):                              # shapes are checked before anything runs
        net(torch.randn(ns + 1, 64, device=DEV), torch.randn(nd, 64, device=DEV), torch.randn(1, 64, device=DEV), graph)
    with torch.no_grad(), pytest.raises(ValueError):
        net(torch.randn(4, 64, device=DEV), torch.randn(3, 64, device=DEV), torch.randn(1, 64, device=DEV), graph)


@pytest.mark.parametrize("C,H,steps,nsteps", [(8, 32, 1, 2), (16, 64, 2, 3), (5, 128, 1, 1)])
def test_forecaster_and_rollout_vs_oracle(ga, C, H, steps, nsteps):
    """grid -> mesh -> grid encode-process-decode on InteractionNet blocks, autoregressive rollout."""
    from gwen_amd import g2m
    from gwen_amd.forecaster import InteractionForecaster, edge_features
    from oracle import interaction_oracle as IO
    m = ga.geodesic_mesh(5)
    torch.manual_seed(SEED)
    model = InteractionForecaster(C, H, steps)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    a, b = g2m.grid_mesh_edges(m)
    cell = m.pos[m.faces].mean(axis=1)
    cell /= np.linalg.norm(cell, axis=1, keepdims=True)
    f = [torch.from_numpy(x).double() for x in (edge_features(cell, m.pos, a), edge_features(m.pos, m.pos, m.edge_index),
                                                 edge_features(m.pos, cell, b))]
    sd = {k: v.double() for k, v in model.state_dict().items()}
    x0 = torch.randn(m.faces.shape[0], C, generator=torch.Generator().manual_seed(SEED))
    want, cur = [], x0.double()
    for _ in range(nsteps):
        cur = IO.forecaster_step(sd, cur, torch.from_numpy(m.pos.astype(np.float32)).double(), torch.from_numpy(a),
                                 torch.from_numpy(m.edge_index), torch.from_numpy(b), *f, steps)
        want.append(cur)
    graphs = InteractionForecaster.prepare(m, DEV)
    model = model.to(DEV)
    got = model.rollout(x0.to(DEV), graphs, nsteps)
    with torch.no_grad():
        one = model(x0.to(DEV), graphs)
    assert torch.equal(one, got[0])
    for g_, w_ in zip(got, want):
        assert rel_err(g_, w_) <= REL_TOL
    replayed = model.rollout(x0.to(DEV), graphs, nsteps, graphed=True)      # one captured step, replayed
    assert all(torch.equal(a, b) for a, b in zip(replayed, got))


def test_ensemble_forecast_single_rank(ga):
    """c5 shape on one rank: all local members through ONE launch set per step (block-diagonal graph),
    gathered once (no process group: the gather is the identity); batched / member-by-member and graphed /
    eager paths agree bitwise."""
    from gwen_amd.forecaster import InteractionForecaster, ensemble_forecast
    m = ga.geodesic_mesh(4)
    torch.manual_seed(SEED)
    model = InteractionForecaster(8, 32, 2).to(DEV).eval()
    graphs = model.prepare(m, DEV)
    xm = torch.randn(3, m.faces.shape[0], 8, device=DEV)
    a = ensemble_forecast(model, graphs, xm, 2, 3, graphed=True)
    b = ensemble_forecast(model, graphs, xm, 2, 3, graphed=False)
    c = ensemble_forecast(model, graphs, xm, 2, 3, graphed=False, batched=False)
    d = ensemble_forecast(model, graphs, xm, 2, 3, graphed=True, batched=False)
    assert a.shape == (3, m.faces.shape[0], 8) and torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    want = torch.stack([model.rollout(xm[i], graphs, 2)[-1] for i in range(3)])
    assert torch.equal(a, want)
    with torch.no_grad():                                         # the members axis of forward()
        one = model(xm, graphs)
        assert torch.equal(one, torch.stack([model(xm[i], graphs) for i in range(3)]))


@pytest.mark.parametrize("F,members", [(64, 4), (128, 3), (256, 2)])
def test_members_axis_block_diagonal_vs_oracle(ga, F, members):
    """[members, ...] through the batched graphs at every K6 width against the fp64 oracle per member."""
    from gwen_amd import g2m
    from gwen_amd.forecaster import InteractionForecaster, edge_features
    from oracle import interaction_oracle as IO
    m = ga.geodesic_mesh(6, reorder="hilbert")
    C, steps = 5, 2
    torch.manual_seed(SEED + F)
    model = InteractionForecaster(C, F, steps)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    a, b = g2m.grid_mesh_edges(m)
    cell = m.pos[m.faces].mean(axis=1)
    cell /= np.linalg.norm(cell, axis=1, keepdims=True)
    f = [torch.from_numpy(v).double() for v in (edge_features(cell, m.pos, a), edge_features(m.pos, m.pos, m.edge_index),
                                                 edge_features(m.pos, cell, b))]
    sd = {k: v.double() for k, v in model.state_dict().items()}
    x0 = torch.randn(members, m.faces.shape[0], C, generator=torch.Generator().manual_seed(SEED))
    graphs = InteractionForecaster.prepare(m, DEV)
    gb = graphs.batched(members)
    assert gb.mesh.num_edges == members * graphs.mesh.num_edges and gb.mesh.num_dst == members * m.num_nodes
    assert graphs.batched(members) is gb                                  # built once
    model = model.to(DEV)
    with torch.no_grad():
        got = model(x0.to(DEV), graphs)
    for k in range(members):
        want = IO.forecaster_step(sd, x0[k].double(), torch.from_numpy(m.pos.astype(np.float32)).double(),
                                  torch.from_numpy(a), torch.from_numpy(m.edge_index), torch.from_numpy(b), *f, steps)
        assert rel_err(got[k], want) <= REL_TOL


@pytest.mark.parametrize("seed", range(16))
def test_interaction_block_on_random_bipartite_graphs(ga, seed):
    """Random shapes: skewed target degrees (empty rows, rows longer than several passes), ragged row
    counts around the 64-row pass, every supported width and activation."""
    from gwen_amd.interaction import InteractionNet, interaction_graph
    from oracle import interaction_oracle as IO
    rng = np.random.default_rng(1000 + seed)
    F = [32, 64, 128, 256][seed % 4]
    act = ["silu", "relu", "none"][seed % 3]
    aggr = ["sum", "mean"][seed % 2]
    ns, nd = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    e = int(rng.integers(1, 3000))
    p = rng.random(nd) ** 4 + 1e-9                       # a few heavy targets, many light or empty ones
    dst = rng.choice(nd, size=e, p=p / p.sum())
    src = rng.integers(0, ns, size=e)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    torch.manual_seed(seed)
    net = InteractionNet(F, act, aggr)
    g = torch.Generator().manual_seed(seed)
    xs, xd, ef = torch.randn(ns, F, generator=g), torch.randn(nd, F, generator=g), torch.randn(e, F, generator=g)
    sd = {k: v.double() for k, v in net.state_dict().items()}
    want_x, want_e = IO.interaction(xs.double(), xd.double(), ef.double(), ei, sd, act, aggr)
    graph = interaction_graph(ei.to(DEV), ns, nd)
    net = net.to(DEV)
    with torch.no_grad():
        got_x, got_e = net(xs.to(DEV), xd.to(DEV), graph.sort_edges(ef.to(DEV)), graph)
    assert rel_err(got_x, want_x) <= REL_TOL
    assert rel_err(graph.unsort_edges(got_e), want_e) <= REL_TOL


@pytest.mark.parametrize("seed", range(6))
def test_act_pair_seg_is_act_pair_then_segment_sum_bitwise(ga, seed):
    """gwen_act_pair_seg_f32 (activation + derivative + the per-target sums of the activated rows in ONE pass over edges
    stored by target) == gwen_act_pair_f32 followed by K2 over the edge-position CSR, bit for bit: skewed target degrees
    (empty targets, targets with hundreds of edges), every activation, the target table read through a row stride."""
    from gwen_amd import interaction as I
    rng = np.random.default_rng(50 + seed)
    F = [32, 64, 128, 256, 8, 64][seed]
    act = ["silu", "relu", "none"][seed % 3]
    ns, nd, e = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 4000))
    p = rng.random(nd) ** 4 + 1e-9
    dst = rng.choice(nd, size=e, p=p / p.sum())
    src = rng.integers(0, ns, size=e)
    graph = I.interaction_graph(torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(DEV), ns, nd)
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(e, F, generator=gen).to(DEV)
    ps = torch.randn(ns, F, generator=gen).to(DEV)
    pall = torch.randn(nd, 3 * F, generator=gen).to(DEV)
    pd = pall[:, F:2 * F]                                                       # a strided view, as in the backward
    h0, d0 = I._act_pair(a.clone(), act, ps, graph.src, pd, graph.dst)
    s0 = I._segsum(graph.segments("dst"), h0, nd)
    h1, d1, s1 = I._act_pair_seg(a.clone(), act, ps, graph.src, pd, graph.rowptr, nd)
    assert torch.equal(h0, h1) and torch.equal(d0, d1) and torch.equal(s0, s1)
    ref = torch.zeros(nd, F, dtype=torch.float64, device=DEV).index_add_(0, graph.dst.long(), h0.double())
    assert rel_err(s1, ref) <= 1e-6


@pytest.mark.parametrize("F,act,aggr,bip", [(32, "silu", "sum", False), (64, "relu", "mean", True),
                                           (128, "silu", "sum", True), (64, "none", "sum", False),
                                           (64, "silu", "sum", True), (256, "silu", "sum", False),
                                           (256, "silu", "mean", True), (256, "none", "sum", True)])
def test_interaction_block_backward_vs_oracle_autograd(ga, F, act, aggr, bip):
    """Training through a block (forward on K6, backward assembled from atomic-free launches of libgwen_hip.so:
    csrc/interact_bwd.hip + K2 / K3 / the gradient reductions; at 64 and 256 channels the edge-level half is ONE launch
    of the row-stationary kernel, gwen_mlp2_bwd_f32 -- at 256 with the weight ring and its hand-counted waits -- and the
    wide weight gradients run on the split contractions): every gradient -- x_src, x_dst, e, the 8 parameters -- against
    torch autograd on the fp64 CPU oracle at 1e-4, and two backward runs bitwise equal.  Edge counts are several passes
    of 128 rows plus a ragged tail (the kernel stores whole passes of the hidden layer into a padded buffer)."""
    from gwen_amd.interaction import InteractionNet, interaction_graph
    from oracle import interaction_oracle as IO
    rng = np.random.default_rng(77 + F)
    ns, nd, e_ = (150, 210, 1300) if bip else (180, 180, 1100)
    src, dst = rng.integers(0, ns, size=e_), rng.integers(0, nd, size=e_)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    torch.manual_seed(SEED + F)
    net = InteractionNet(F, act, aggr)
    g = torch.Generator().manual_seed(SEED)
    xs, xd, ef = torch.randn(ns, F, generator=g), torch.randn(nd, F, generator=g), torch.randn(e_, F, generator=g)
    gxo, geo = torch.randn(nd, F, generator=g), torch.randn(e_, F, generator=g)
    sd = {k: v.double().clone().requires_grad_() for k, v in net.state_dict().items()}
    xs64, xd64, ef64 = xs.double().requires_grad_(), xd.double().requires_grad_(), ef.double().requires_grad_()
    wx, we = IO.interaction(xd64 if not bip else xs64, xd64, ef64, ei, sd, act, aggr)
    (wx * gxo.double()).sum().add((we * geo.double()).sum()).backward()
    graph = interaction_graph(ei.to(DEV), ns, nd)
    net = net.to(DEV)
    xsd, xdd = xs.to(DEV).requires_grad_(), xd.to(DEV).requires_grad_()
    efd = graph.sort_edges(ef.to(DEV)).detach().requires_grad_()
    gx, ge = net(xdd if not bip else xsd, xdd, efd, graph)
    ((gx * gxo.to(DEV)).sum() + (ge * graph.sort_edges(geo.to(DEV))).sum()).backward()
    assert rel_err(gx.detach(), wx.detach()) <= REL_TOL
    tol = REL_TOL
    assert rel_err(xdd.grad, xd64.grad) <= tol
    if bip:
        assert rel_err(xsd.grad, xs64.grad) <= tol
    assert rel_err(graph.unsort_edges(efd.grad), ef64.grad) <= tol
    for k, p in net.named_parameters():
        assert rel_err(p.grad, sd[k].grad) <= tol, k
    # fixed-order sums, no atomics: a second run gives the same bits
    first = [t.grad.clone() for t in ([xdd, efd] + ([xsd] if bip else []) + list(net.parameters()))]
    for t in [xdd, efd, xsd] + list(net.parameters()):
        t.grad = None
    gx2, ge2 = net(xdd if not bip else xsd, xdd, efd, graph)
    ((gx2 * gxo.to(DEV)).sum() + (ge2 * graph.sort_edges(geo.to(DEV))).sum()).backward()
    again = [t.grad for t in ([xdd, efd] + ([xsd] if bip else []) + list(net.parameters()))]
    assert all(torch.equal(a, b) for a, b in zip(first, again))


def test_interaction_block_backward_without_edge_update(ga):
    """Encoder / decoder blocks return no edge state (update_edges=False): the message gradient then comes from
    the aggregate alone."""
    from gwen_amd.interaction import InteractionNet, interaction_graph
    from oracle import interaction_oracle as IO
    rng = np.random.default_rng(5)
    F, ns, nd, e_ = 64, 90, 140, 700
    ei = torch.from_numpy(np.stack([rng.integers(0, ns, size=e_), rng.integers(0, nd, size=e_)]).astype(np.int64))
    torch.manual_seed(SEED)
    net = InteractionNet(F, "silu", "sum")
    g = torch.Generator().manual_seed(SEED)
    xs, xd, ef = torch.randn(ns, F, generator=g), torch.randn(nd, F, generator=g), torch.randn(e_, F, generator=g)
    gxo = torch.randn(nd, F, generator=g)
    sd = {k: v.double().clone().requires_grad_() for k, v in net.state_dict().items()}
    xs64, xd64, ef64 = xs.double().requires_grad_(), xd.double().requires_grad_(), ef.double().requires_grad_()
    wx, _ = IO.interaction(xs64, xd64, ef64, ei, sd, "silu", "sum")
    (wx * gxo.double()).sum().backward()
    graph = interaction_graph(ei.to(DEV), ns, nd)
    net = net.to(DEV)
    xsd, xdd = xs.to(DEV).requires_grad_(), xd.to(DEV).requires_grad_()
    efd = graph.sort_edges(ef.to(DEV)).detach().requires_grad_()
    gx, ge = net(xsd, xdd, efd, graph, update_edges=False)
    assert ge.numel() == 0
    (gx * gxo.to(DEV)).sum().backward()
    assert rel_err(xdd.grad, xd64.grad) <= REL_TOL and rel_err(xsd.grad, xs64.grad) <= REL_TOL
    assert rel_err(graph.unsort_edges(efd.grad), ef64.grad) <= REL_TOL
    for k, p in net.named_parameters():
        assert rel_err(p.grad, sd[k].grad) <= REL_TOL, k


def test_forecaster_training_step(ga):
    """The c5-shaped model trains: gradients reach every parameter and Adam lowers the loss."""
    from gwen_amd.forecaster import InteractionForecaster
    m = ga.geodesic_mesh(4, reorder="hilbert")
    torch.manual_seed(SEED)
    model = InteractionForecaster(6, 32, 2).to(DEV)
    graphs = model.prepare(m, DEV)
    x = torch.randn(m.faces.shape[0], 6, device=DEV)
    y = torch.randn(m.faces.shape[0], 6, device=DEV)
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = (model(x, graphs) - y).square().mean()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    with torch.no_grad():                                       # inference path unchanged by training mode
        a = model(x, graphs)
    assert torch.isfinite(a).all()
