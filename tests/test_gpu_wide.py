"""K8 (tile-staged wide layer, csrc/wide.hip) and its tile prep (csrc/tiles.hip) on the device.

Checked against: a numpy restatement of the tile layout, K4 bit for bit (K8 is K4's 3xbf16 arithmetic term
for term), the C oracle (oracle/gcn_ref.c, fp64) within tolerance, and at config size (nu = 100, 256
channels, 4 members) through linearity / determinism / member independence plus the oracle on one member.
The layer being replaced: torch-geometric GCNConv.forward as called at
/root/reference/src/gwen/models_gnn.py:147-149,:204-206."""
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


def _mesh_graph(ga, nu, reorder="morton"):
    m = ga.geodesic_mesh(nu, reorder=reorder)
    ei = torch.from_numpy(m.edge_index)
    return m, ei, ga.prepare_graph(ei.to(DEV), m.num_nodes)


@pytest.mark.parametrize("nu,reorder", [(4, None), (10, "morton"), (30, "morton"), (30, "hilbert"), (30, None)])
def test_tile_layout_matches_numpy(ga, nu, reorder):
    m, ei, g = _mesh_graph(ga, nu, reorder)
    tiles = g.tiles()
    if reorder is None and nu >= 30:
        assert tiles is None       # generator order has no locality: 64 rows name > 192 distinct sources
        return
    assert tiles is not None
    t_rows, t_lid, t_val = (t.cpu().numpy() for t in tiles[:3])
    umax = 0
    t_lid = t_lid.view(np.uint16)
    n = m.num_nodes
    rowptr, col, val = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy()
    T = (n + 63) // 64
    assert t_rows.shape == (T * 192,) and t_lid.shape == (T * 512,) and t_val.shape == (T * 512,)
    for t in range(T):
        ent_c = np.full(512, -1, dtype=np.int64)
        ent_w = np.zeros(512, dtype=np.float32)
        for lr in range(64):
            r = t * 64 + lr
            if r >= n:
                continue
            a, b = rowptr[r], rowptr[r + 1]
            assert b - a <= 8
            ent_c[lr * 8: lr * 8 + b - a] = col[a:b]
            ent_w[lr * 8: lr * 8 + b - a] = val[a:b]
        uniq = np.unique(ent_c[ent_c >= 0])
        nu_ = len(uniq)
        assert nu_ <= 192
        umax = max(umax, nu_)
        slots = t_rows[t * 192:(t + 1) * 192]
        assert np.array_equal(slots[:nu_], uniq)
        for k in range(nu_, 192):
            assert slots[k] == (uniq[0] if (k & ~3) < nu_ else -1)
        assert np.array_equal(t_val[t * 512:(t + 1) * 512], ent_w)
        lid = t_lid[t * 512:(t + 1) * 512].astype(np.int64)
        for s in range(512):
            c = ent_c[s] if ent_c[s] >= 0 else ent_c[s & ~7]
            if c >= 0:
                assert uniq[lid[s]] == c
            else:
                assert lid[s] == 0
    assert tiles[3] == umax


def test_untileable_graphs_report_none(ga):
    ei = torch.from_numpy(ga.complete_graph(40)).to(DEV)            # rows of 40 entries
    assert ga.prepare_graph(ei, 40).tiles() is None
    # bounded degree but no locality: 64 rows name ~448 distinct sources
    g = torch.Generator().manual_seed(SEED)
    n = 20000
    src = torch.randint(0, n, (6 * n,), generator=g)
    dst = torch.arange(n).repeat(6)
    assert ga.prepare_graph(torch.stack([src, dst]).to(DEV), n).tiles() is None
    with pytest.raises(ValueError):
        from gwen_amd import ops
        ops.wide_layer(ga.prepare_graph(ei, 40), torch.zeros(40, 64, device=DEV), torch.zeros(64, 64, device=DEV))


@pytest.mark.parametrize("fin,fout", [(64, 64), (64, 128), (64, 256), (128, 64), (128, 128), (128, 256),
                                      (256, 64), (256, 128), (256, 256)])
@pytest.mark.parametrize("nu,reorder", [(4, "morton"), (13, "morton"), (13, "hilbert")])
def test_wide_equals_k4_bitwise_and_oracle(ga, cref, fin, fout, nu, reorder):
    """morton at nu = 13 has unions above 128 (one chunk of DMA in flight), hilbert stays below (two)."""
    from gwen_amd import ops
    m, ei, g = _mesh_graph(ga, nu, reorder)
    n = m.num_nodes
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(SEED + fin))
    w, b = make_params(fin, fout)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    for relu in (False, True):
        got = ops.wide_layer(g, xd, wd, bd, relu=relu, contract="3xbf16")
        k4 = ops.layer_fused(g, xd, wd, bd, relu=relu, exact=False)
        assert torch.equal(got, k4), (fin, fout, relu)
        ref = cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=relu, f64=True)
        assert rel_err(got, ref) <= 2e-5
    assert torch.equal(ops.wide_layer(g, xd, wd, None, contract="3xbf16"), ops.layer_fused(g, xd, wd, None))
    # bf16x6 on K8: graphs whose unions stay within 128 rows; K4's bf16x6 bit for bit
    umax = g.tiles()[3]
    if umax <= 128:            # (256 -> 256 runs as two 256 -> 128 launches: W's three images for 128 columns fit)
        got6 = ops.wide_layer(g, xd, wd, bd, relu=True, contract="bf16x6")
        assert torch.equal(got6, ops.layer_fused(g, xd, wd, bd, relu=True, contract="bf16x6")), (fin, fout)
        ref = cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
        assert rel_err(got6, ref) <= 2e-6
    else:
        from gwen_amd._lib import GwenHipError
        with pytest.raises(GwenHipError):
            ops.wide_layer(g, xd, wd, bd, relu=True, contract="bf16x6")


def test_wide_members_axis_and_few_tiles(ga, cref):
    """[members, N, F] shares one tile layout; fewer tiles than CUs and more tiles than CUs."""
    from gwen_amd import ops
    for nu, members, reorder in ((3, 3, "morton"), (20, 5, "morton"), (20, 3, "hilbert")):
        m, ei, g = _mesh_graph(ga, nu, reorder)
        n = m.num_nodes
        x = torch.randn(members, n, 128, generator=torch.Generator().manual_seed(SEED))
        w, b = make_params(128, 128)
        got = ops.wide_layer(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
        for k in range(members):
            one = ops.wide_layer(g, x[k].to(DEV), w.to(DEV), b.to(DEV), relu=True)
            assert torch.equal(got[k], one)
        ref = cref.conv(x[members - 1].numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
        assert rel_err(got[members - 1], ref) <= 2e-5


@pytest.mark.parametrize("contract", ["3xbf16", "bf16x6"])
@pytest.mark.parametrize("fout", [128, 256])
@pytest.mark.parametrize("nu,members", [(20, 3), (40, 3), (40, 7)])
def test_wide_256_skewed_row_tiles_all_members_bitwise(ga, nu, members, fout, contract):
    """256 input channels with unions <= 128 rows run the SKEWED schedule (row tile t of a tile t steps behind row tile
    0, one row tile stored per step, three drain steps per block): 1, 2-3 and 6-7 tiles per block, block ranges that
    cross member boundaries and the partial last tile of every member -- every member bitwise K4's result.  bf16x6
    (256 -> 256: two 256 -> 128 launches writing column halves through the output's row stride) likewise."""
    from gwen_amd import ops
    m, ei, g = _mesh_graph(ga, nu, "hilbert")
    n = m.num_nodes
    assert g.tiles()[3] <= 128 and n % 64 != 0
    x = torch.randn(members, n, 256, generator=torch.Generator().manual_seed(SEED + nu)).to(DEV)
    w, b = make_params(256, fout)
    wd, bd = w.to(DEV), b.to(DEV)
    for relu in (True, False):
        got = ops.wide_layer(g, x, wd, bd, relu=relu, contract=contract)
        for k in range(members):
            assert torch.equal(got[k], ops.layer_fused(g, x[k], wd, bd, relu=relu, contract=contract)), (k, relu)
    again = ops.wide_layer(g, x, wd, bd, relu=False, contract=contract)
    assert torch.equal(again, got)


def test_wide_bipartite(ga):
    """Rectangular graphs (grid -> mesh maps of SURVEY 8(f) f2) tile the same way: x has N_src rows."""
    from gwen_amd import ops
    from gwen_amd.g2m import grid_mesh_edges
    from gwen_amd.graph import prepare_bipartite
    m = ga.geodesic_mesh(12, reorder="morton")
    g2m_e, m2g_e = grid_mesh_edges(m)
    n_mesh, n_grid = m.num_nodes, m.faces.shape[0]
    g2m = prepare_bipartite(torch.from_numpy(g2m_e).to(DEV), n_grid, n_mesh)
    m2g = prepare_bipartite(torch.from_numpy(m2g_e).to(DEV), n_mesh, n_grid)
    tiled = 0
    for g, n_src in ((g2m, n_grid), (m2g, n_mesh)):
        if g.tiles() is None:
            continue
        tiled += 1
        x = torch.randn(n_src, 64, device=DEV)
        w, b = make_params(64, 128)
        got = ops.wide_layer(g, x, w.to(DEV), b.to(DEV), relu=True, contract="3xbf16")
        assert torch.equal(got, ops.layer_fused(g, x, w.to(DEV), b.to(DEV), relu=True))
        # the default precision (f16x3 on K8) on a rectangular graph: fp32-class, i.e. within 2e-6 of K4's bf16x6
        got16 = ops.wide_layer(g, x, w.to(DEV), b.to(DEV), relu=True)
        assert rel_err(got16, ops.layer_fused(g, x, w.to(DEV), b.to(DEV), relu=True, contract="bf16x6")) <= 2e-6
        assert torch.equal(got16, ops.wide_layer(g, x, w.to(DEV), b.to(DEV), relu=True, contract="f16x3"))
    assert tiled >= 1


@pytest.mark.parametrize("contract,tol", [("3xbf16", 2e-5), ("bf16x6", 2e-6)])
def test_c3_layer_at_config_size_four_members(ga, cref, contract, tol):
    """BASELINE c3 / c5 per-GPU load: nu = 100 (N = 100 002, E = 600 000), 256 -> 256, 4 members (the
    HBM-bound leg of bench.py: its 3xbf16 tier and its bf16x6 two-launch form; the default f16x3 has the same test in
    tests/test_gpu_f16x3.py).  One member against the C oracle (fp64); all members through determinism, K4 equality,
    equality with the one-member launch and linearity in x."""
    from gwen_amd import ops
    m, ei, g = _mesh_graph(ga, 100, "hilbert")
    n = m.num_nodes
    assert (n, m.num_edges) == (100002, 600000) and g.tiles()[3] <= 128
    members = 4
    gen = torch.Generator().manual_seed(SEED)
    x = torch.randn(members, n, 256, generator=gen)
    w, b = make_params(256, 256)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    got = ops.wide_layer(g, xd, wd, bd, relu=True, contract=contract)
    assert torch.equal(got, ops.wide_layer(g, xd, wd, bd, relu=True, contract=contract))
    assert torch.equal(got, ops.layer_fused(g, xd, wd, bd, relu=True, contract=contract))
    for k in (0, 3):
        assert torch.equal(got[k], ops.wide_layer(g, xd[k], wd, bd, relu=True, contract=contract))
    ref = cref.conv(x[2].numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
    assert rel_err(got[2], ref) <= tol
    # per-row check as well: no row may hide behind the tensor's largest value
    diff = (got[2].cpu().double() - torch.from_numpy(ref)).abs().amax(1)
    scale = torch.from_numpy(ref).abs().amax(1).clamp_min(1e-3)
    assert float((diff / scale).max()) <= 1e-3
    # linearity (no bias, no ReLU): f(x0 + 2 x1) = f(x0) + 2 f(x1)
    lin = ops.wide_layer(g, xd[0] + 2 * xd[1], wd, None, contract=contract)
    parts = ops.wide_layer(g, xd[0], wd, None, contract=contract) + 2 * ops.wide_layer(g, xd[1], wd, None, contract=contract)
    assert rel_err(lin, parts) <= tol


def test_clustered_order_gives_k8_on_unordered_meshes(ga, cref):
    """A caller's arbitrary node numbering (here: a random relabelling of the nu = 40 mesh -- what an unordered
    ``edge_index`` looks like) does not tile; ``GraphCSR.clustered`` grows a locality order itself and the stack
    launcher runs K8 in it, output rows back in the caller's order and bitwise equal to K4 on the caller's graph
    (VERDICT r2 item 8).  Graphs without any locality (random sources) and hub rows still report None."""
    m = ga.geodesic_mesh(40)                                         # lexicographic generator order
    n = m.num_nodes
    rng = np.random.default_rng(SEED)
    relabel = rng.permutation(n)
    ei = torch.from_numpy(relabel[m.edge_index]).contiguous()       # same mesh, shuffled node ids
    g = ga.prepare_graph(ei.to(DEV), n)
    assert g.tiles() is None                                         # 64 consecutive ids are scattered rows
    cl = g.clustered()
    assert cl is not None
    perm, inv, gp = cl
    assert sorted(perm.tolist()) == list(range(n)) and torch.equal(inv[perm], torch.arange(n, device=DEV))
    umax = gp.tiles()[3]
    assert umax <= 192
    members, F = 4, 128
    torch.manual_seed(SEED)
    params = [make_params(F, F, seed=SEED + k) for k in range(2)]
    x = torch.randn(members, n, F, generator=torch.Generator().manual_seed(SEED))
    for order in ("auto_x3", "auto_x6", "auto"):
        if order == "auto_x6" and umax > 128:
            continue                                                 # bf16x6 on K8 needs unions within 128 rows
        layers = [(w.to(DEV), b.to(DEV), True, order) for w, b in params]
        plan = ga.StackForward(layers, g)
        ev = ga.KernelEvents(8)
        got = plan.run(x.to(DEV), events=ev)
        assert [k for k, *_ in ev.durations()] == ["wide", "wide"]
        if order == "auto":            # the default: f16x3 on K8 (unions up to 192 rows too) -- fp32-class, not K4's bits
            got_default = got
            continue
        k4 = ga.StackForward([(w, b, r, "fused" if order == "auto_x6" else "fused_x3") for w, b, r, _ in layers], g)
        ev4 = ga.KernelEvents(8)
        want = k4.run(x.to(DEV), events=ev4)
        assert [k for k, *_ in ev4.durations()] == ["layer", "layer"]
        assert torch.equal(got, want)                                # same terms, same order: same bits
        out = torch.empty_like(got)
        assert plan.run(x.to(DEV), out=out) is out and torch.equal(out, want)
    ref = x[1].numpy()
    for w, b in params:
        ref = cref.conv(ref.astype(np.float32), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
    assert rel_err(want[1], ref) <= 2e-5
    assert rel_err(got_default[1], ref) <= 2e-6
    # no locality to find / hubs: still None, the planner stays on K4
    gen = torch.Generator().manual_seed(SEED)
    nn_ = 20000
    rnd = torch.stack([torch.randint(0, nn_, (6 * nn_,), generator=gen), torch.arange(nn_).repeat(6)])
    assert ga.prepare_graph(rnd.to(DEV), nn_).clustered() is None
    assert ga.prepare_graph(torch.from_numpy(ga.complete_graph(200)).to(DEV), 200).clustered() is None
