"""Pins the oracle: closed-form known-answer tests and the cross-check of the
two independent restatements (torch op sequence vs plain C).  CPU only.

PARITY UNPINNED at the reference level: the reference's tests hold no numeric vector for this path
(/root/reference/tests/test_gwen/test_models.py:19,36 mock the layers out) and torch-geometric is
not importable here; these KATs are what anchors the oracle instead.
"""
import numpy as np
import pytest
import torch

from helpers import SEED, graph_cases, make_params, random_multigraph, rel_err
from oracle import gcn_oracle as O

CASES = graph_cases()
IDS = [c[0] for c in CASES]


def test_kat_complete_graph_mean():
    # the reference's own graph family: erdos_renyi_graph(n, 1) -- utils.py:176
    from gwen_amd.mesh import complete_graph
    for n in (2, 7, 125):
        ei = torch.from_numpy(complete_graph(n))
        x = torch.randn(n, 5, generator=torch.Generator().manual_seed(SEED)).double()
        w, b = make_params(5, 3)
        out = O.gcn_conv(x, ei, w.double(), b.double())
        want = (x @ w.double().t()).mean(0, keepdim=True) + b.double()
        assert rel_err(out, want.expand_as(out)) < 1e-12
        _, wn = O.gcn_norm(ei, None, n)
        assert torch.allclose(wn, torch.full_like(wn, 1.0 / n), rtol=1e-6)


def test_kat_path_graph_weights():
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    e2, w = O.gcn_norm(ei, None, 3, dtype=torch.float64)
    # deg = (2, 3, 2): w01 = 1/sqrt(6), loops 1/2, 1/3, 1/2
    r6 = 1 / 6 ** 0.5
    assert e2.tolist() == [[0, 1, 1, 2, 0, 1, 2], [1, 0, 2, 1, 0, 1, 2]]
    assert torch.allclose(w, torch.tensor([r6, r6, r6, r6, 0.5, 1 / 3, 0.5], dtype=torch.float64))


def test_kat_directed_cycle():
    n = 6
    ei = torch.stack([torch.arange(n), (torch.arange(n) + 1) % n])
    x = torch.randn(n, 4, generator=torch.Generator().manual_seed(SEED)).double()
    out = O.gcn_conv(x, ei, torch.eye(4, dtype=torch.float64), None)
    assert rel_err(out, 0.5 * (x + torch.roll(x, 1, 0))) < 1e-12


def test_kat_empty_edges_is_linear():
    x = torch.randn(5, 3).double()
    w, b = make_params(3, 4)
    out = O.gcn_conv(x, torch.zeros(2, 0, dtype=torch.long), w.double(), b.double())
    assert rel_err(out, x @ w.double().t() + b.double()) < 1e-12


def test_self_loops_and_duplicates():
    # explicit loops are replaced by ONE unit loop; parallel non-loop edges both count
    ei = torch.tensor([[0, 0, 1, 1, 1], [0, 0, 0, 0, 1]])     # two loops at 0, 1->0 twice, loop at 1
    e2, w = O.gcn_norm(ei, None, 2, dtype=torch.float64)
    assert e2.tolist() == [[1, 1, 0, 1], [0, 0, 0, 1]]
    # deg(0) = 2 + 1 = 3, deg(1) = 1
    assert torch.allclose(w, torch.tensor([1 / 3 ** 0.5, 1 / 3 ** 0.5, 1 / 3, 1.0], dtype=torch.float64))
    # weighted: an existing loop keeps its weight, the LAST duplicate wins; others get fill
    ew = torch.tensor([5.0, 7.0, 1.0, 1.0, 3.0], dtype=torch.float64)
    e3, _ = O.add_remaining_self_loops(ei, ew, 1.0, 2)
    _, w3 = O.add_remaining_self_loops(ei, ew, 1.0, 2)
    assert w3.tolist() == [1.0, 1.0, 7.0, 3.0]
    _, w4 = O.add_remaining_self_loops(torch.tensor([[0], [1]]), torch.tensor([2.0]), 2.0, 3)
    assert w4.tolist() == [2.0, 2.0, 2.0, 2.0]                # improved=True fill


def test_isolated_node_without_loops():
    ei = torch.tensor([[0], [1]])
    _, w = O.gcn_norm(ei, None, 3, add_self_loops=False)
    assert w.tolist() == [0.0]                                # deg(0)=0 -> inf -> 0


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_permutation_equivariance_and_linearity(case):
    name, n, ei = case
    gen = torch.Generator().manual_seed(SEED)
    x, y = torch.randn(n, 6, generator=gen).double(), torch.randn(n, 6, generator=gen).double()
    w, b = make_params(6, 4)
    w, b = w.double(), b.double()
    f = lambda t, e=ei: O.gcn_conv(t, e, w, b)
    assert rel_err(f(2 * x - y), 2 * (f(x) - b) - (f(y) - b) + b) < 1e-12
    perm = torch.randperm(n, generator=gen)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(n)
    assert rel_err(O.gcn_conv(x[perm], inv[ei], w, b), f(x)[perm]) < 1e-12


# ---- the two restatements agree --------------------------------------------------------------
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_torch_and_c_oracles_agree_bitwise(cref, case, weighted):
    name, n, ei = case
    ew = None
    if weighted:
        ew = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(SEED)) + 0.25
    e2, w = O.gcn_norm(ei, ew, n)
    s, d, wc = cref.norm(ei.numpy(), None if ew is None else ew.numpy(), n)
    np.testing.assert_array_equal(e2[0].numpy(), s)
    np.testing.assert_array_equal(e2[1].numpy(), d)
    assert w.numpy().tobytes() == wc.tobytes()
    h = torch.randn(n, 9, generator=torch.Generator().manual_seed(SEED))
    got = O.propagate(h, e2, w, n).numpy()
    assert got.tobytes() == cref.propagate(s, d, wc, h.numpy()).tobytes()


def test_c_layer_close_to_torch_layer_and_f64(cref):
    name, n, ei = CASES[0]
    x = torch.randn(n, 16, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(16, 8)
    t32 = O.gcn_conv(x, ei, w, b)
    c32 = cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy())
    c64 = cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), f64=True)
    t64 = O.gcn_conv(x.double(), ei, w.double(), b.double())
    assert rel_err(c64, t64) < 1e-12
    assert rel_err(c32, t64) < 1e-6 and rel_err(t32, t64) < 1e-6
    r = cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True)
    assert (r >= 0).all() and np.array_equal(r, np.maximum(c32, 0))


def test_c_oracle_rejects_bad_index(cref):
    with pytest.raises(IndexError):
        cref.norm(np.array([[0, 3], [1, 0]]), None, 3)


def test_model_composition_matches_manual_chain():
    # models_gnn.py:147-149 (conv1..3 + ReLU), :204-206 (upconv3,4 + ReLU, upconv5 plain)
    name, n, ei = CASES[0]
    torch.manual_seed(SEED)
    m = O.OracleGNNModel(O.OracleGNNConfig(n, n, 8, 8, 16))
    x = torch.randn(n, 8)
    sd = m.state_dict()
    t = x
    for i, (key, act) in enumerate([("down_conv_layers.conv1", 1), ("down_conv_layers.conv2", 1),
                                    ("down_conv_layers.conv3", 1), ("up_conv_layers.upconv3", 1),
                                    ("up_conv_layers.upconv4", 1), ("up_conv_layers.upconv5", 0)]):
        t = O.gcn_conv(t, ei, sd[f"conv_layers.{key}.lin.weight"], sd[f"conv_layers.{key}.bias"])
        if act:
            t = torch.relu(t)
    assert torch.equal(t, m(x, ei))
    assert len(sd) == 20


def test_dense_matrix_form_of_the_published_rule():
    """A third, independent statement of the rule the oracle restates -- Kipf & Welling's
    X' = D^-1/2 (A + I) D^-1/2 X W^T + b with A_ij = number of edges j -> i -- as dense fp64 matrix algebra (numpy),
    against the op-sequence oracle on simple graphs, multigraphs (parallel edges count twice) and graphs that already
    carry self-loops (de-duplicated to one loop of weight 1).  Anchors the oracle to the published formula rather
    than to its own implementation (the reference holds no numeric fixture: SURVEY 8c)."""
    import numpy as np
    from helpers import random_multigraph
    from gwen_amd.mesh import geodesic_mesh
    rng = np.random.default_rng(5)
    cases = [("mesh", geodesic_mesh(3).num_nodes, torch.from_numpy(geodesic_mesh(3).edge_index)),
             ("multi", 60, random_multigraph(60, 400, self_loops=9, dup=30, isolate=3)),
             ("path", 3, torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]]))]
    for name, n, ei in cases:
        fin, fout = 5, 7
        x = rng.standard_normal((n, fin))
        w = rng.standard_normal((fout, fin))
        b = rng.standard_normal(fout)
        a = np.zeros((n, n))
        for s, d in ei.t().tolist():
            if s != d:
                a[d, s] += 1.0                      # parallel edges add up; explicit loops are dropped ...
        a += np.eye(n)                              # ... and every node gets exactly one loop of weight 1
        deg = a.sum(axis=1)                         # in-degree over targets, loop included
        dis = 1.0 / np.sqrt(deg)
        want = (dis[:, None] * a * dis[None, :]) @ x @ w.T + b
        got = O.gcn_conv(torch.from_numpy(x), ei, torch.from_numpy(w), torch.from_numpy(b)).numpy()
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12), name
