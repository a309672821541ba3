"""K9 (csrc/hops.hip): chains of narrow GCN layers in one launch by overlapped tiling, and the hop layout.
Against a numpy restatement of the layout and the C oracle (oracle/gcn_ref.c, fp64) layer by layer.  The
layers: GNNModel's middle gathers, /root/reference/src/gwen/models_gnn.py:148-149, :204."""
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


@pytest.mark.parametrize("nu,H", [(5, 3), (12, 3), (12, 4), (20, 2)])
def test_hop_layout_matches_numpy(ga, nu, H):
    m = ga.geodesic_mesh(nu, reorder="hilbert")
    n = m.num_nodes
    g = ga.prepare_graph(torch.from_numpy(m.edge_index).to(DEV), n)
    hops = g.hops(H)
    if hops is None:
        pytest.skip("hop sets exceed the kernel's budget at this size / depth")
    h_cnt, h_rows, h_lid, h_val, hh = hops
    h_cnt, h_rows = h_cnt.cpu().numpy().reshape(-1, 6), h_rows.cpu().numpy().reshape(-1, 288)
    h_lid = h_lid.cpu().numpy().view(np.uint16).reshape(-1, 192, 8)
    h_val = h_val.cpu().numpy().reshape(-1, 192, 8)
    rowptr, col, val = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy()
    for t in range((n + 63) // 64):
        own = list(range(t * 64, min(n, t * 64 + 64)))
        levels, seen, frontier = [own], set(own), own
        for _ in range(H):
            new = sorted({int(c) for r in frontier for c in col[rowptr[r]:rowptr[r + 1]]} - seen)
            levels.append(new); seen |= set(new); frontier = new
        want = [r for lv in levels for r in lv]
        counts = np.cumsum([len(lv) for lv in levels])
        assert list(h_cnt[t, :H + 1]) == list(counts)
        assert list(h_rows[t, :counts[-1]]) == want and (h_rows[t, counts[-1]:] == -1).all()
        for r in range(counts[H - 1]):
            row = want[r]
            a, b = rowptr[row], rowptr[row + 1]
            for e in range(8):
                if a + e < b:
                    assert want[h_lid[t, r, e]] == col[a + e] and h_val[t, r, e] == val[a + e]
                else:
                    assert h_val[t, r, e] == 0.0 and want[h_lid[t, r, e]] == col[a]


def _ref_chain(cref, x, ei, stages):
    cur = x.numpy().astype(np.float64)
    n = cur.shape[0]
    for w, b, relu in stages:
        f = cur.shape[1]
        wn = np.eye(f, dtype=np.float32) if w is None else w.numpy()
        cur = cref.conv(cur.astype(np.float32), ei.numpy(), wn, None if b is None else b.numpy(), relu=relu, f64=True)
    return cur


@pytest.mark.parametrize("widths,pre", [((32, 16, 32), True), ((32, 16, 32, 32), True), ((16, 16), False),
                                        ((32, 32, 16, 32), False), ((8, 16, 4), False)])
@pytest.mark.parametrize("nu,members", [(6, 1), (60, 3)])
def test_narrow_chain_vs_oracle(ga, cref, widths, pre, nu, members):
    """pre: the first stage takes an already projected input (no weight).  The c2 model's fused middle is
    ((32, 16, 32[, 64]), pre)."""
    from gwen_amd import ops
    m = ga.geodesic_mesh(nu, reorder="hilbert")
    n, ei = m.num_nodes, torch.from_numpy(m.edge_index)
    g = ga.prepare_graph(ei.to(DEV), n)
    gen = torch.Generator().manual_seed(SEED)
    stages, f = [], widths[0]
    if pre:
        stages.append((None, torch.randn(f, generator=gen) * 0.1, True))
    for k, fo in enumerate(widths[1:]):
        w, b = make_params(f, fo, seed=SEED + k)
        stages.append((w, b, k + 2 < len(widths)))
        f = fo
    if len(stages) > 3 and g.hops(4) is None:
        pytest.skip("4-hop sets exceed the budget")
    x = torch.randn(members, n, widths[0], generator=gen)
    dev_stages = [(None if w is None else w.to(DEV), None if b is None else b.to(DEV), r) for w, b, r in stages]
    got = ops.narrow_chain(g, x.to(DEV) if members > 1 else x[0].to(DEV), dev_stages)
    got = got.view(members, n, -1)
    assert torch.equal(got, ops.narrow_chain(g, x.to(DEV), dev_stages).view(members, n, -1))
    for k in range(members):
        want = _ref_chain(cref, x[k], ei, stages)
        assert rel_err(got[k], want) <= 2e-6, (widths, k)          # fp32 FMAs: as close as the exact path


def test_narrow_chain_rejects_what_it_cannot_do(ga):
    from gwen_amd import ops
    m = ga.geodesic_mesh(6, reorder="hilbert")
    g = ga.prepare_graph(torch.from_numpy(m.edge_index).to(DEV), m.num_nodes)
    x = torch.randn(m.num_nodes, 64, device=DEV)
    with pytest.raises(ValueError):
        ops.narrow_chain(g, x, [(torch.randn(32, 64, device=DEV), None, True)])          # fin 64 > 32
    with pytest.raises(ValueError):
        ops.narrow_chain(g, x[:, :8].contiguous(), [(torch.randn(24, 8, device=DEV), None, True)])   # 24: not 2^k
    kn = ga.prepare_graph(torch.from_numpy(ga.complete_graph(300)).to(DEV), 300)
    assert kn.hops(3) is None
