"""K7 (gwen_gcn_small_layer_f32, gwen_gcn_dense_f32): whole GCNConv layers on graphs of at most 256 nodes
with wide features -- the reference's own shape (complete graph over ~125 members,
/root/reference/src/gwen/utils.py:175-176; hidden 1024, config.json:12) -- against the CPU oracle."""
import pytest
import torch

from helpers import REL_TOL, SEED, graph_cases, make_params, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
def _cases():
    from gwen_amd.mesh import complete_graph
    cases = [c for c in graph_cases() if c[1] <= 256]
    cases.append(("K150", 150, torch.from_numpy(complete_graph(150))))      # the reference's upper range
    cases.append(("K256", 256, torch.from_numpy(complete_graph(256))))
    from helpers import random_multigraph
    cases.append(("multi100", 100, random_multigraph(100, 700, self_loops=15, dup=60, isolate=5)))
    cases.append(("multi250", 250, random_multigraph(250, 3000, seed=7, self_loops=30, dup=200, isolate=9)))
    return cases


SMALL = _cases()


@pytest.fixture(scope="module")
def ga(hip_lib):
    import gwen_amd
    return gwen_amd


@pytest.mark.parametrize("name,n,ei", SMALL, ids=[c[0] for c in SMALL])
def test_dense_matrix_is_the_normalised_adjacency(ga, name, n, ei):
    from oracle import gcn_oracle as O
    g = ga.prepare_graph(ei.to(DEV), n)
    npad = 128 if n <= 128 else 256
    d = g.dense().cpu().view(npad, npad)
    ei2, w = O.gcn_norm(ei, None, n)
    want = torch.zeros(npad, npad, dtype=torch.float64)
    want.index_put_((ei2[1], ei2[0]), w.double(), accumulate=True)
    assert rel_err(d, want) <= 1e-6
    assert torch.count_nonzero(d[n:]) == 0 and torch.count_nonzero(d[:, n:]) == 0


@pytest.mark.parametrize("name,n,ei", SMALL, ids=[c[0] for c in SMALL])
@pytest.mark.parametrize("fin,fout", [(32, 16), (64, 64), (1024, 512), (4096, 64), (8192, 128), (64, 4096)])
@pytest.mark.parametrize("members,relu,use_bias", [(1, True, True), (3, False, False)])
def test_small_layer_vs_oracle(ga, name, n, ei, fin, fout, members, relu, use_bias):
    from gwen_amd import ops
    from oracle import gcn_oracle as O
    w, b = make_params(fin, fout)
    g = torch.Generator().manual_seed(SEED + fin)
    x = torch.randn(members, n, fin, generator=g)
    want = torch.stack([O.gcn_conv(x[m].double(), ei, w.double(), b.double() if use_bias else None)
                        for m in range(members)])
    if relu:
        want = torch.relu(want)
    graph = ga.prepare_graph(ei.to(DEV), n)
    xs = x.to(DEV) if members > 1 else x[0].to(DEV)
    got = ops.small_layer(graph, xs, w.to(DEV), b.to(DEV) if use_bias else None, relu)
    from gwen_amd.forward import pack_weight
    img = pack_weight(w.to(DEV))                 # same hi/lo values, streamed in fragment order
    again = ops.small_layer(graph, xs, w.to(DEV), b.to(DEV) if use_bias else None, relu, packed=img)
    assert rel_err(got.view(members, n, fout), want) <= 2e-5
    assert img.numel() == w.numel() * 4 and torch.equal(got, again)
    # bf16x6 (the default precision of the host API): fp32-class; the packed images are bf16x3's and are not used
    got6 = ops.small_layer(graph, xs, w.to(DEV), b.to(DEV) if use_bias else None, relu, contract="bf16x6")
    e6, e3 = rel_err(got6.view(members, n, fout), want), rel_err(got.view(members, n, fout), want)
    assert e6 <= 5e-6 and e6 <= e3 + 1e-7          # fp32 accumulation over K up to 8192 is what is left
    assert torch.equal(got6, ops.small_layer(graph, xs, w.to(DEV), b.to(DEV) if use_bias else None, relu, packed=img,
                                             contract="bf16x6"))


@pytest.mark.parametrize("n,c,h", [(125, 2048, 256), (150, 4096, 512)])
def test_reference_shaped_model_runs_on_k7(ga, n, c, h):
    """GNNModel on the reference's graph family: every layer is one K7 launch, result within tolerance."""
    from gwen_amd.mesh import complete_graph
    from oracle import gcn_oracle as O
    ei = torch.from_numpy(complete_graph(n))
    torch.manual_seed(SEED)
    model = ga.GNNModel(ga.GNNConfig(n, n, c, c, h))
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    oracle = O.OracleGNNModel(O.OracleGNNConfig(n, n, c, c, h)).double()
    oracle.load_state_dict({k: v.double() for k, v in model.state_dict().items()}, strict=True)
    x = torch.randn(n, c, generator=torch.Generator().manual_seed(SEED))
    with torch.no_grad():
        want = oracle(x.double(), ei)
    model = model.to(DEV).eval()
    graph = model.prepare(ei.to(DEV), n)
    plan = ga.StackForward(model.stack(), graph)
    ev = ga.KernelEvents(12)
    got = plan.run(x.to(DEV), events=ev)
    kinds = [k for k, *_ in ev.durations()]
    assert kinds == ["small"] * 6
    assert rel_err(got, want) <= REL_TOL
    with torch.no_grad():
        assert torch.equal(model(x.to(DEV), ei.to(DEV)), got)


def test_small_layer_rejects_what_it_cannot_do(ga):
    from gwen_amd import ops
    from helpers import random_multigraph
    big = ga.prepare_graph(random_multigraph(300, 2000).to(DEV), 300)
    assert big.dense() is None
    with pytest.raises(ValueError):
        ops.small_layer(big, torch.randn(300, 64, device=DEV), torch.randn(64, 64, device=DEV))
    g = ga.prepare_graph(torch.tensor([[0, 1], [1, 0]], device=DEV), 2)
    with pytest.raises(ValueError):
        ops.small_layer(g, torch.randn(2, 24, device=DEV), torch.randn(16, 24, device=DEV))


def test_training_step_through_k7_matches_the_oracle(ga):
    """One layer + ReLU on the reference's graph family: forward on K7, backward on K2^T / K3 / grad
    kernels; gradients against the oracle's autograd in fp64."""
    from gwen_amd.mesh import complete_graph
    from oracle import gcn_oracle as O
    n, fin, fout = 125, 64, 32
    ei = torch.from_numpy(complete_graph(n))
    torch.manual_seed(SEED)
    conv = ga.GCNConv(fin, fout)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    ref = O.OracleGCNConv(fin, fout).double()
    ref.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(SEED))
    xr = x.double().requires_grad_(True)
    torch.relu(ref(xr, ei)).square().sum().backward()
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = conv(xg, ei.to(DEV), relu=True)
    out.square().sum().backward()
    assert rel_err(out.detach(), torch.relu(ref(xr, ei)).detach()) <= 2e-5
    assert rel_err(xg.grad, xr.grad) <= REL_TOL
    assert rel_err(conv.lin.weight.grad, ref.lin.weight.grad) <= REL_TOL
    assert rel_err(conv.bias.grad, ref.bias.grad) <= REL_TOL


@pytest.mark.parametrize("name,n,ei", [c for c in SMALL if c[1] >= 2], ids=[c[0] for c in SMALL if c[1] >= 2])
@pytest.mark.parametrize("f", [256, 1024])
def test_backward_propagate_as_one_dense_product(ga, name, n, ei, f):
    """ops._propagate_transposed: on square graphs of at most 256 nodes with wide rows the backward's gh = A~^T g is ONE
    dense product D^T g (K3 with g as the [K, N] operand, ops.linear_nn) -- against K2 over the transposed CSR (the exact,
    sequential sum) and the fp64 oracle's A~^T g; narrow rows and the fp32 precision stay on K2 (bitwise)."""
    from gwen_amd import ops
    from oracle import gcn_oracle as O
    g = ga.prepare_graph(ei.to(DEV), n)
    grad = torch.randn(n, f, generator=torch.Generator().manual_seed(SEED + n)).to(DEV)
    exact = ops.propagate(g, grad, transposed=True)
    for contract, tol in (("bf16x6", 2e-6), ("f16x3", 2e-6), ("3xbf16", 3e-5)):
        got = ops._propagate_transposed(g, grad, contract)
        assert rel_err(got, exact) <= tol, (contract, rel_err(got, exact))
        assert torch.equal(got, ops._propagate_transposed(g, grad, contract))
    assert torch.equal(ops._propagate_transposed(g, grad, "fp32"), exact)
    assert torch.equal(ops._propagate_transposed(g, grad[:, :64].contiguous(), "bf16x6"),
                       ops.propagate(g, grad[:, :64].contiguous(), transposed=True))
    dense = g.dense_transposed_square()
    assert dense is not None and dense.shape == (n, n)
    want = dense.double() @ grad.double()
    assert rel_err(exact, want) <= 1e-6


def test_weighted_small_graph(ga):
    """Explicit edge weights (and improved=True, which only acts on weighted graphs) reach K7 through
    the dense matrix."""
    from gwen_amd import ops
    from helpers import random_multigraph
    from oracle import gcn_oracle as O
    n, fin, fout = 90, 64, 48
    ei = random_multigraph(n, 500, seed=3, self_loops=10, dup=40)
    gen = torch.Generator().manual_seed(SEED)
    ew = torch.rand(ei.size(1), generator=gen) + 0.1
    w, b = make_params(fin, fout)
    x = torch.randn(n, fin, generator=gen)
    for improved in (False, True):
        want = O.gcn_conv(x.double(), ei, w.double(), b.double(), ew.double(), improved=improved)
        g = ga.prepare_graph(ei.to(DEV), n, ew.to(DEV), improved=improved)
        got = ops.small_layer(g, x.to(DEV), w.to(DEV), b.to(DEV))
        assert rel_err(got, want) <= 2e-5


@pytest.mark.parametrize("seed", range(12))
def test_small_layer_on_random_graphs_and_shapes(ga, seed):
    """Random node counts up to 256, random multigraphs, random (Fin, Fout) among the supported multiples."""
    import numpy as np
    from gwen_amd import ops
    from gwen_amd.forward import pack_weight
    from helpers import random_multigraph
    from oracle import gcn_oracle as O
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(1, 257))
    fin = int(rng.integers(1, 40)) * 32
    fout = int(rng.integers(1, 40)) * 16
    e = int(rng.integers(0, 6 * n + 1))
    ei = random_multigraph(n, e, seed=seed, self_loops=int(rng.integers(0, 5)), dup=int(rng.integers(0, 9)) if e else 0)
    w, b = make_params(fin, fout, seed=seed)
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(seed))
    want = torch.relu(O.gcn_conv(x.double(), ei, w.double(), b.double()))
    g = ga.prepare_graph(ei.to(DEV), n)
    got = ops.small_layer(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
    packed = ops.small_layer(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True, packed=pack_weight(w.to(DEV)))
    assert rel_err(got, want) <= 2e-5 and torch.equal(got, packed)
