"""GCNConv module surface on the device: every constructor option of the reference's layer class
(torch-geometric 2.3.1 GCNConv; the reference passes none of them, models_gnn.py:118-184, but the drop-in
keeps them), edge weights, input validation, caching behaviour."""
import numpy as np
import pytest
import torch

from helpers import REL_TOL, SEED, graph_cases, make_params, random_multigraph, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = graph_cases()


@pytest.fixture(scope="module")
def ga(hip_lib):
    import gwen_amd
    return gwen_amd


@pytest.mark.parametrize("improved,loops,normalize,bias", [(True, True, True, True), (False, False, True, True),
                                                            (False, True, False, True), (False, True, True, False),
                                                            (True, False, False, False)])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("fin,fout", [(64, 32), (10, 6)])
def test_constructor_options_match_oracle(ga, improved, loops, normalize, bias, weighted, fin, fout):
    from oracle import gcn_oracle as O
    n, ei = 300, random_multigraph(300, 2000, self_loops=40, dup=100, isolate=7)
    gen = torch.Generator().manual_seed(SEED)
    x = torch.randn(n, fin, generator=gen)
    ew = (torch.rand(ei.size(1), generator=gen) + 0.25) if weighted else None
    w, b = make_params(fin, fout)
    conv = ga.GCNConv(fin, fout, improved=improved, add_self_loops=loops, normalize=normalize, bias=bias).to(DEV)
    with torch.no_grad():
        conv.lin.weight.copy_(w)
        if bias:
            conv.bias.copy_(b)
        got = conv(x.to(DEV), ei.to(DEV), None if ew is None else ew.to(DEV)).cpu()
    ref = O.gcn_conv(x.double(), ei, w.double(), b.double() if bias else None,
                     None if ew is None else ew.double(), improved=improved, add_self_loops=loops,
                     normalize=normalize)
    assert rel_err(got, ref) <= REL_TOL


def test_cached_layer_pins_its_graph_and_reset_drops_it(ga):
    n, ei = CASES[0][1], CASES[0][2].to(DEV)
    conv = ga.GCNConv(8, 8, cached=True).to(DEV)
    x = torch.randn(n, 8, device=DEV)
    y1 = conv(x, ei)
    assert conv._cached_graph is not None
    other = torch.stack([ei[1], ei[0]])               # a different tensor object: still the pinned graph
    assert torch.equal(conv(x, other), y1)
    conv.reset_parameters()
    assert conv._cached_graph is None


def test_input_validation_on_device(ga):
    n, ei = CASES[0][1], CASES[0][2].to(DEV)
    conv = ga.GCNConv(8, 8).to(DEV)
    with pytest.raises(TypeError):
        conv(torch.randn(n, 8, device=DEV, dtype=torch.float64), ei)
    with pytest.raises(TypeError):
        conv(torch.randn(n, 8, device=DEV).half(), ei)
    with pytest.raises(ValueError):
        conv(torch.randn(n + 1, 8, device=DEV), ga.prepare_graph(ei, n))      # graph/rows mismatch
    with pytest.raises(RuntimeError):
        conv(torch.randn(n, 8), ei)                                            # CPU x, no fallback
    with pytest.raises(ValueError):
        ga.prepare_graph(ei, n, torch.ones(3, device=DEV))                     # weight length
    with pytest.raises(IndexError):
        conv(torch.randn(5, 8, device=DEV), ei)                                # indices >= N


def test_non_contiguous_and_strided_inputs(ga):
    from oracle import gcn_oracle as O
    n, ei = CASES[0][1], CASES[0][2]
    w, b = make_params(16, 16)
    conv = ga.GCNConv(16, 16).to(DEV)
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
    big = torch.randn(n, 40, generator=torch.Generator().manual_seed(SEED))
    x = big[:, 3:35:2]                                       # strided view, 16 columns
    with torch.no_grad():
        got = conv(x.to(DEV)[:, :], ei.to(DEV)).cpu()
        got2 = conv(big.to(DEV)[:, 3:35:2], ei.to(DEV)).cpu()
    ref = O.gcn_conv(x.contiguous(), ei, w, b)
    assert rel_err(got, ref) <= REL_TOL and torch.equal(got, got2)


def test_forward_does_not_touch_its_inputs(ga):
    n, ei = CASES[0][1], CASES[0][2].to(DEV)
    x = torch.randn(n, 64, device=DEV)
    x0, e0 = x.clone(), ei.clone()
    model = ga.GNNModel(ga.GNNConfig(n, n, 64, 64, 64)).to(DEV).eval()
    with torch.no_grad():
        model(x, ei)
    assert torch.equal(x, x0) and torch.equal(ei, e0)


def test_reference_scale_workload(ga):
    """The reference's own shape: K_125 member graph, wide flattened fields, hidden_feats 1024
    (config.json:9,12) -- widths 3000 -> 1024 -> 512 -> 256 -> 512 -> 1024 -> 3000."""
    from oracle import gcn_oracle as O
    n, c, h = 125, 3000, 1024
    ei = torch.from_numpy(ga.complete_graph(n))
    torch.manual_seed(SEED)
    ref = O.OracleGNNModel(O.OracleGNNConfig(n, n, c, c, h))
    model = ga.GNNModel(ga.GNNConfig(n, n, c, c, h))
    model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(DEV).eval()
    x = torch.randn(n, c)
    with torch.no_grad():
        got = model(x.to(DEV), ei.to(DEV)).cpu()
        want = ref(x, ei)
    assert rel_err(got, want) <= REL_TOL
    # on K_N every output row is the same vector
    assert rel_err(got, got[0:1].expand_as(got)) <= 1e-5


@pytest.mark.parametrize("F,steps", [(256, 4), (64, 3)])
def test_c3_processor_stack_vs_oracle(F, steps):
    """BASELINE config c3: chained F -> F GCN layers with ReLU on one mesh (oracle.processor_stack)."""
    import gwen_amd
    from oracle import gcn_oracle as O
    mesh = gwen_amd.geodesic_mesh(10)
    ei = torch.from_numpy(mesh.edge_index)
    ws, bs = zip(*[make_params(F, F, seed=SEED + k) for k in range(steps)])
    x = torch.randn(mesh.num_nodes, F, generator=torch.Generator().manual_seed(SEED))
    want = O.processor_stack(x.double(), ei, [w.double() for w in ws], [b.double() for b in bs])
    g = gwen_amd.prepare_graph(ei.to(DEV), mesh.num_nodes)
    plan = gwen_amd.StackForward([(w.to(DEV), b.to(DEV), True, "auto") for w, b in zip(ws, bs)], g)
    assert rel_err(plan.run(x.to(DEV)), want) <= REL_TOL


def test_checksum_is_a_function_of_the_bytes(ga):
    """gwen_checksum128: equal bytes -> equal key (whatever the tensor object), any changed byte, a swapped
    pair of words or a different length -> another key; ragged byte counts and empty buffers included."""
    from gwen_amd.graph import content_key
    g = torch.Generator().manual_seed(SEED)
    for n in (0, 1, 7, 4096, 1_200_001):
        a = torch.randint(-2 ** 40, 2 ** 40, (n,), generator=g).to(DEV)
        k = content_key(a)
        assert content_key(a.clone()) == k and content_key(a) == k
        if n >= 2:
            b = a.clone(); b[n // 2] += 1
            assert content_key(b) != k
            c = a.clone(); c[0], c[n - 1] = a[n - 1], a[0]
            if not torch.equal(c, a):
                assert content_key(c) != k                      # position-dependent
            assert content_key(a[:-1]) != k
    u8 = torch.arange(13, dtype=torch.uint8, device=DEV)        # 13 bytes: one word + a 5-byte tail
    k8 = content_key(u8)
    v8 = u8.clone(); v8[12] = 99
    assert content_key(u8.clone()) == k8 and content_key(v8) != k8


def test_fresh_edge_index_tensors_hit_the_cache_by_content(ga):
    """The reference's loops hand every forward a new edge_index tensor with the same edges
    (models_gnn.py:351-360): K1 runs once, relabelled edges run it again, results are unchanged."""
    n = 125
    ei_host = torch.from_numpy(ga.complete_graph(n))
    conv = ga.GCNConv(32, 16).to(DEV)
    x = torch.randn(n, 32, device=DEV)
    cache = ga.default_cache()
    cache.clear()
    m0, c0 = cache.misses, cache.content_hits
    with torch.no_grad():
        y0 = conv(x, ei_host.to(DEV))
        for _ in range(3):
            assert torch.equal(conv(x, ei_host.to(DEV)), y0)
    assert (cache.misses - m0, cache.content_hits - c0) == (1, 3)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(SEED))
    relabelled = perm[ei_host]                                    # same graph family, different bytes
    with torch.no_grad():
        y1 = conv(x[torch.argsort(perm)], relabelled.to(DEV))
    assert cache.misses - m0 == 2
    assert rel_err(y1[perm], y0) <= 1e-5                          # K_N relabelled is K_N: same rows, permuted
    with torch.inference_mode():
        ei_inf = ei_host.to(DEV)
        assert torch.equal(conv(x, ei_inf), y0) and cache.misses - m0 == 2
