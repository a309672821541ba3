"""Host-side logic on CPU: mesh generator, module surface / state_dict, pickling, error behaviour,
graph cache and ensemble partitioning."""
import pickle

import numpy as np
import pytest
import torch

import gwen_amd
from gwen_amd import ensemble, graph as G
from helpers import SEED
from oracle import gcn_oracle as O


@pytest.mark.parametrize("nu", [1, 2, 3, 7, 10])
def test_geodesic_mesh_counts_and_conventions(nu):
    m = gwen_amd.geodesic_mesh(nu)
    n, e = 10 * nu * nu + 2, 60 * nu * nu
    assert m.num_nodes == n and m.num_edges == e and len(m.faces) == 20 * nu * nu
    ei = m.edge_index
    assert ei.dtype == np.int64 and ei.shape == (2, e)
    assert (ei[0] != ei[1]).all()                                   # no self-loops
    key = ei[0] * n + ei[1]
    assert (np.diff(key) > 0).all()                                 # sorted by (row, col), no duplicates
    rev = np.sort(ei[1] * n + ei[0])
    assert np.array_equal(rev, key)                                 # both directions present
    deg = np.bincount(ei[1], minlength=n)
    assert (deg == 5).sum() == 12 and (deg == 6).sum() == n - 12 if nu > 1 else (deg == 5).all()
    assert np.allclose(np.linalg.norm(m.pos, axis=1), 1.0)
    # Euler: V - E/2 + F = 2
    assert n - e // 2 + len(m.faces) == 2


@pytest.mark.parametrize("reorder", ["morton", "hilbert"])
def test_c2_mesh_size_and_relabelling_is_isomorphic(reorder):
    m = gwen_amd.geodesic_mesh(100)
    assert (m.num_nodes, m.num_edges) == (100002, 600000)
    a, b = gwen_amd.geodesic_mesh(6), gwen_amd.geodesic_mesh(6, reorder=reorder)
    inv = np.empty_like(b.perm); inv[b.perm] = np.arange(len(inv))
    ea = inv[a.edge_index]
    ka = np.sort(ea[0] * a.num_nodes + ea[1]); kb = b.edge_index[0] * b.num_nodes + b.edge_index[1]
    assert np.array_equal(ka, kb)
    assert np.allclose(a.pos[b.perm], b.pos)


def test_hilbert_order_keeps_tile_unions_small():
    """What K8 needs of the node order (csrc/tiles.hip): 64 consecutive destination rows of the nu = 100
    mesh name at most 128 distinct source rows (self-loops included) under the Hilbert relabelling."""
    m = gwen_amd.geodesic_mesh(100, reorder="hilbert")
    n = m.num_nodes
    src = np.concatenate([m.edge_index[0], np.arange(n)])
    dst = np.concatenate([m.edge_index[1], np.arange(n)])
    cnt = np.bincount(np.unique((dst // 64) * n + src) // n)
    assert cnt.max() <= 128 and cnt.mean() < 110


def test_complete_graph_matches_reference_producer_conventions():
    ei = gwen_amd.complete_graph(5)       # erdos_renyi_graph(5, 1): utils.py:176
    assert ei.shape == (2, 20) and (ei[0] != ei[1]).all()
    assert (np.diff(ei[0] * 5 + ei[1]) > 0).all()


def test_state_dict_is_appendix_b():
    c, h = 12, 32
    m = gwen_amd.GNNModel(gwen_amd.GNNConfig(9, 9, c, c + 1, h))
    sd = m.state_dict()
    want = {}
    dims = {"down_conv_layers.conv1": (h, c), "down_conv_layers.conv2": (h // 2, h),
            "down_conv_layers.conv3": (h // 4, h // 2), "down_conv_layers.conv4": (h // 8, h // 4),
            "down_conv_layers.conv5": (h // 16, h // 8), "up_conv_layers.upconv1": (h // 8, h // 16),
            "up_conv_layers.upconv2": (h // 4, h // 8), "up_conv_layers.upconv3": (h // 2, h // 4),
            "up_conv_layers.upconv4": (h, h // 2), "up_conv_layers.upconv5": (c + 1, h)}
    for k, (o, i) in dims.items():
        want[f"conv_layers.{k}.bias"] = (o,)
        want[f"conv_layers.{k}.lin.weight"] = (o, i)
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert list(sd)[:2] == ["conv_layers.down_conv_layers.conv1.bias",
                            "conv_layers.down_conv_layers.conv1.lin.weight"]   # bias first, as PyG


def test_state_dict_round_trip_with_oracle_model_strict():
    cfg = (7, 7, 8, 8, 16)
    torch.manual_seed(SEED)
    ref = O.OracleGNNModel(O.OracleGNNConfig(*cfg))
    mine = gwen_amd.GNNModel(gwen_amd.GNNConfig(*cfg))
    mine.load_state_dict(ref.state_dict(), strict=True)
    ref2 = O.OracleGNNModel(O.OracleGNNConfig(*cfg))
    ref2.load_state_dict(mine.state_dict(), strict=True)
    for a, b in zip(ref.state_dict().values(), ref2.state_dict().values()):
        assert torch.equal(a, b)


def test_initialisation_glorot_and_zero_bias():
    torch.manual_seed(SEED)
    conv = gwen_amd.GCNConv(300, 200)
    a = (6.0 / 500) ** 0.5
    w = conv.lin.weight
    assert float(w.detach().abs().max()) <= a and float(w.detach().abs().max()) > 0.95 * a
    assert abs(float(w.detach().mean())) < 0.01 * a * 10 and torch.count_nonzero(conv.bias) == 0
    assert gwen_amd.GCNConv(4, 4, bias=False).bias is None
    assert list(gwen_amd.GCNConv(4, 4, bias=False).state_dict()) == ["lin.weight"]
    with pytest.raises(TypeError):
        gwen_amd.GCNConv(4, 4, flow="target_to_source")


def test_hidden_feats_8_gives_zero_width_layers_like_the_reference():
    # BASELINE config c1 uses 8 hidden channels: conv5 = GCNConv(1, 0), upconv1 = GCNConv(0, 1)
    m = gwen_amd.GNNModel(gwen_amd.GNNConfig(1, 1, 8, 8, 8))
    assert tuple(m.conv_layers.down_conv_layers.conv5.lin.weight.shape) == (0, 1)
    assert tuple(m.conv_layers.up_conv_layers.upconv1.lin.weight.shape) == (1, 0)


def test_module_is_picklable_and_drops_device_state():
    m = gwen_amd.GNNModel(gwen_amd.GNNConfig(3, 3, 8, 8, 16))
    m.conv_layers.down_conv_layers.conv1._cached_graph = object()
    m2 = pickle.loads(pickle.dumps(m))
    assert m2.conv_layers.down_conv_layers.conv1._cached_graph is None
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    cfg = pickle.loads(pickle.dumps(gwen_amd.GNNConfig(3, 3, 8, 8, 16)))
    assert cfg.hidden_feats == 16


def test_cpu_tensors_raise_runtime_error_not_fallback():
    m = gwen_amd.GNNModel(gwen_amd.GNNConfig(3, 3, 8, 8, 16))
    x, ei = torch.randn(3, 8), torch.tensor([[0, 1], [1, 2]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, ei)
    conv = gwen_amd.GCNConv(8, 4)
    with pytest.raises(RuntimeError):
        conv(x, ei)
    with pytest.raises(ValueError):
        conv(torch.randn(3, 5), ei)
    with pytest.raises(ValueError):
        conv(torch.randn(8), ei)
    with pytest.raises(TypeError):
        conv(np.zeros((3, 8)), ei)


def test_graph_cache_identity_then_content(monkeypatch):
    """Identity first (free), content second (128-bit checksum): a NEW tensor with the same edges -- what
    every NeighborLoader batch of the reference is (models_gnn.py:351-360) -- re-uses the prepared graph;
    different bytes, options or an in-place edit never do.  (CPU stand-ins for K1 and the checksum.)"""
    calls = []
    monkeypatch.setattr(G, "prepare_graph", lambda ei, n, ew=None, **kw: calls.append((id(ei), n, kw)) or object())
    monkeypatch.setattr(G, "content_key", lambda t: (hash(t.numpy().tobytes()), tuple(t.shape).__hash__()))
    monkeypatch.setattr(G.GraphCache, "_check", staticmethod(lambda ei, n: None))
    cache = G.GraphCache(capacity=2)
    opts = dict(add_self_loops=True, improved=False, normalize=True)
    a = torch.tensor([[0, 1], [1, 0]])
    g1 = cache.get(a, 2, None, **opts)
    assert cache.get(a, 2, None, **opts) is g1 and len(calls) == 1        # same object: identity hit
    assert (cache.hits, cache.content_hits, cache.misses) == (1, 0, 1)
    b = a.clone()
    assert cache.get(b, 2, None, **opts) is g1 and len(calls) == 1        # equal content, new object: content hit
    assert cache.content_hits == 1
    assert cache.get(b, 2, None, **opts) is g1 and cache.hits == 2        # ... and known by identity from now on
    a[0, 0] = 1                                                           # in-place edit bumps _version
    assert cache.get(a, 2, None, **opts) is not g1 and len(calls) == 2    # other bytes: miss
    assert cache.get(a, 3, None, **opts) is not None and len(calls) == 3  # other num_nodes: miss
    w = torch.ones(2)
    gw = cache.get(a, 2, w, **opts)
    assert cache.get(a, 2, w, **opts) is gw and len(calls) == 4
    assert cache.get(a, 2, w.clone(), **opts) is gw and len(calls) == 4   # equal weights by content
    assert cache.get(a, 2, 2 * w, **opts) is not gw and len(calls) == 5
    assert len(cache._by_content) <= 2                                    # LRU bound
    # a dead tensor's identity entry disappears with it (weakref callback); its id may be reused safely
    c = torch.tensor([[0], [1]]); cache.get(c, 2, None, **opts); key_id = id(c)
    assert key_id in cache._by_id
    del c
    assert key_id not in cache._by_id
    n_before = len(calls)
    d = torch.tensor([[1], [0]])
    cache.get(d, 2, None, **opts)
    assert len(calls) == n_before + 1


def test_graph_cache_handles_inference_mode_tensors(monkeypatch):
    """Tensors created under torch.inference_mode() have no version counter: never trusted by identity."""
    calls = []
    monkeypatch.setattr(G, "prepare_graph", lambda ei, n, ew=None, **kw: calls.append(1) or object())
    monkeypatch.setattr(G, "content_key", lambda t: (hash(t.numpy().tobytes()), 0))
    monkeypatch.setattr(G.GraphCache, "_check", staticmethod(lambda ei, n: None))
    cache = G.GraphCache()
    opts = dict(add_self_loops=True, improved=False, normalize=True)
    with torch.inference_mode():
        a = torch.tensor([[0, 1], [1, 0]])
        g1 = cache.get(a, 2, None, **opts)
        assert cache.get(a, 2, None, **opts) is g1 and len(calls) == 1 and cache.hits == 0 and cache.content_hits == 1

def test_member_range_partitions_every_member_once():
    for members in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [ensemble.member_range(members, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == members
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == ensemble.member_counts(members, world)
    with pytest.raises(ValueError):
        ensemble.member_range(4, 4, 4)


def test_gather_members_single_process_is_identity():
    x = torch.randn(3, 5, 2)
    assert ensemble.gather_members(x, 3) is x
    with pytest.raises(ValueError):
        ensemble.gather_members(x, 4)
    out = ensemble.ensemble_rollout(lambda s: s + 1, x, 4, 3)
    assert torch.allclose(out, x + 4)


def test_edge_features_of_the_forecaster_graphs():
    """[length, displacement] per edge: reversing an edge flips the displacement and keeps the length;
    grid -> mesh edges link every cell with its three corners (host-side geometry only)."""
    import numpy as np
    from gwen_amd.forecaster import edge_features
    from gwen_amd.g2m import grid_mesh_edges
    from gwen_amd.mesh import geodesic_mesh
    m = geodesic_mesh(3)
    f = edge_features(m.pos, m.pos, m.edge_index)
    assert f.shape == (m.num_edges, 4) and f.dtype == np.float32
    assert np.allclose(f[:, 0], np.linalg.norm(f[:, 1:], axis=1), atol=1e-6)
    rev = edge_features(m.pos, m.pos, m.edge_index[::-1])
    assert np.allclose(rev[:, 0], f[:, 0]) and np.allclose(rev[:, 1:], -f[:, 1:])
    g2m, m2g = grid_mesh_edges(m)
    cell = m.pos[m.faces].mean(axis=1)
    a, b = edge_features(cell, m.pos, g2m), edge_features(m.pos, cell, m2g)
    assert np.allclose(a[:, 0], b[:, 0]) and np.allclose(a[:, 1:], -b[:, 1:], atol=1e-6)
    assert a[:, 0].max() < 2.0 / 3 + 0.5                       # a corner is close to its cell centre


def test_loss_func_is_the_reference_masked_l1():
    """gwen_amd.loss_func (masked sum, no boolean indexing) == the reference's
    l1_loss(output[mask], target[mask]) (models_gnn.py:261-265) in value and gradient; an empty mask gives NaN
    in both."""
    torch.manual_seed(3)
    o = torch.randn(50, 7, requires_grad=True)
    t = torch.randn(50, 7)
    m = torch.rand(50) < 0.4
    a = gwen_amd.loss_func(o, t, m)
    b = torch.nn.functional.l1_loss(o[m], t[m])
    assert abs(float(a) - float(b)) <= 1e-6
    ga, = torch.autograd.grad(a, o)
    gb, = torch.autograd.grad(b, o)
    assert torch.allclose(ga, gb, atol=1e-8)
    o3, t3 = torch.randn(3, 50, 7), torch.randn(3, 50, 7)
    assert abs(float(gwen_amd.loss_func(o3, t3, m)) - float(torch.nn.functional.l1_loss(o3[:, m], t3[:, m]))) <= 1e-6
    none = torch.zeros(50, dtype=torch.bool)
    assert torch.isnan(gwen_amd.loss_func(o, t, none)) and torch.isnan(torch.nn.functional.l1_loss(o[none], t[none]))


def test_loss_func_index_mask_is_the_reference_expression():
    """An integer INDEX mask selects (and may repeat) rows in the reference's ``output[target_mask]``
    (models_gnn.py:261-265); only bool masks take the masked-sum path."""
    torch.manual_seed(4)
    o = torch.randn(20, 8, requires_grad=True)
    t = torch.randn(20, 8)
    idx = torch.tensor([3, 3, 7, 19, 0])                       # repeats count twice in the reference
    a = gwen_amd.loss_func(o, t, idx)
    b = torch.nn.functional.l1_loss(o[idx], t[idx])
    assert float(a) == float(b)
    ga, = torch.autograd.grad(a, o)
    gb, = torch.autograd.grad(b, o)
    assert torch.equal(ga, gb)
    # a 0/1 integer vector of length N is an index tensor too (rows 0 and 1), not a row selector
    z = torch.zeros(20, dtype=torch.long); z[5] = 1
    assert float(gwen_amd.loss_func(o, t, z)) == float(torch.nn.functional.l1_loss(o[z], t[z]))


def test_cluster_rows64_host_grows_compact_patches():
    """gwen_cluster_rows64_host (csrc/cluster.hip) is host code: no GPU needed.  On a randomly relabelled geodesic
    mesh (an unordered edge_index) it returns a permutation whose consecutive blocks of 64 rows name few distinct
    sources -- what K8's tile layout needs (<= 192) -- where the caller's own numbering names hundreds."""
    import ctypes as C
    from gwen_amd import _lib, build
    from gwen_amd.mesh import geodesic_mesh
    build.build()
    L = _lib.lib()
    m = geodesic_mesh(30)
    n = m.num_nodes
    rng = np.random.default_rng(3)
    relabel = rng.permutation(n)
    ei = relabel[m.edge_index]
    # CSR by target with the self-loop last, as K1 stores it
    order = np.argsort(ei[1], kind="stable")
    src, dst = ei[0][order], ei[1][order]
    counts = np.bincount(dst, minlength=n) + 1
    rowptr = np.zeros(n + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(counts)
    col = np.empty(int(rowptr[-1]), dtype=np.int32)
    fill = rowptr[:-1].copy()
    for s_, d_ in zip(src, dst):
        col[fill[d_]] = s_
        fill[d_] += 1
    col[fill] = np.arange(n, dtype=np.int32)                     # the completed self-loops

    def worst_union(perm):
        worst = 0
        for t in range(0, n, 64):
            rows = perm[t:t + 64]
            srcs = np.unique(np.concatenate([col[rowptr[r]:rowptr[r + 1]] for r in rows]))
            worst = max(worst, len(srcs))
        return worst

    perm = np.empty(n, dtype=np.int32)
    rc = L.gwen_cluster_rows64_host(rowptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), n, n,
                                    perm.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert np.array_equal(np.sort(perm), np.arange(n))           # a permutation
    assert worst_union(np.arange(n)) > 192                       # the caller's numbering does not tile
    assert worst_union(perm) <= 192                              # the grown order does
    # rectangular graphs are refused, an empty graph is fine
    assert L.gwen_cluster_rows64_host(rowptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), n, n + 1,
                                      perm.ctypes.data_as(C.c_void_p)) != 0
    assert L.gwen_cluster_rows64_host(None, None, 0, 0, None) == 0


def test_precision_names_and_what_each_kernel_runs():
    """The library default is "f16x3": fp32-class on the kernel's own split (K8: two scaled fp16 images; every other
    kernel: bf16x6).  The names map to layer orders and contraction codes without a GPU (host logic only)."""
    import gwen_amd
    from gwen_amd import _lib, ops
    from gwen_amd.forward import _ORDERS
    conv = gwen_amd.GCNConv(64, 64)
    assert conv.order == "auto" and conv.precision == "f16x3"
    for name, order in (("f16x3", "auto"), ("bf16x6", "auto_x6"), ("3xbf16", "auto_x3")):
        conv.precision = name
        assert conv.order == order and conv.precision == name
        assert ops.contract_of_order(order) == name and order in ops.AUTO_ORDERS
    conv.precision = "fp32"
    assert conv.order == "fused_exact" and conv.precision == "fp32"
    with pytest.raises(ValueError):
        conv.precision = "bf16"
    assert _ORDERS["auto"] == (_lib.ORDER_AUTO, _lib.CONTRACT_F16X3)
    assert _ORDERS["auto_x6"] == (_lib.ORDER_AUTO, _lib.CONTRACT_BF16X6)
    assert _ORDERS["fused"] == (_lib.ORDER_FUSED, _lib.CONTRACT_BF16X6)
    # K3 / K4 / K5 / K7 run bf16x6 for an f16x3 layer; the other codes pass through
    assert _lib.dense_contract(_lib.CONTRACT_F16X3) == _lib.CONTRACT_BF16X6
    for code in (_lib.CONTRACT_BF16X3, _lib.CONTRACT_F32, _lib.CONTRACT_BF16X6):
        assert _lib.dense_contract(code) == code
    assert ops._dense_code("f16x3") == _lib.CONTRACT_BF16X6 and ops._contract_code("f16x3") == _lib.CONTRACT_F16X3
    # K8 has f16x3 at every width pair it supports (a library query: no GPU work)
    lib = _lib.lib()
    for fin in (64, 128, 256):
        for fout in (64, 128, 256):
            assert lib.gwen_gcn_wide_contract_supported(fin, fout, _lib.CONTRACT_F16X3) == 1
            assert _lib.wide_contract(fin, fout, _lib.CONTRACT_F16X3) == _lib.CONTRACT_F16X3
    assert lib.gwen_gcn_wide_contract_supported(96, 64, _lib.CONTRACT_F16X3) == 0
    model = gwen_amd.GNNModel(gwen_amd.GNNConfig(10, 10, 16, 16, 32))
    assert {c.precision for c in model.modules() if isinstance(c, gwen_amd.GCNConv)} == {"f16x3"}
    model.set_precision("3xbf16")
    assert {c.order for c in model.modules() if isinstance(c, gwen_amd.GCNConv)} == {"auto_x3"}
