"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerance: BASELINE.json north_star states 1e-4 relative fp32 (helpers.REL_TOL).  K1 and K2 are
held to a stricter bar -- BIT-EXACT against the plain-C oracle -- because they reproduce the
reference CPU path's visiting order and rounding (see DESIGN.md "Numerics").
"""
import numpy as np
import pytest
import torch

from helpers import (REL_TOL, SEED, csr_from_oracle, graph_cases, make_params, random_multigraph,
                     rel_err)

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ga(hip_lib):
    import gwen_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gwen_amd


CASES = graph_cases()
IDS = [c[0] for c in CASES]


# ------------------------------------------------------------------------------------------------
# K1: prepared CSR == oracle gcn_norm, entry for entry, bit for bit
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_prep_matches_oracle(ga, cref, case, weighted):
    name, n, ei = case
    ew = None
    if weighted:
        ew = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(SEED)) + 0.25
    g = ga.prepare_graph(ei.to(DEV), n, None if ew is None else ew.to(DEV))
    s, d, w = cref.norm(ei.numpy(), None if ew is None else ew.numpy(), n)
    rowptr, col, val = csr_from_oracle(s, d, w, n)
    nnz = g.nnz()
    assert nnz == len(w)
    np.testing.assert_array_equal(g.rowptr.cpu().numpy(), rowptr)
    np.testing.assert_array_equal(g.col.cpu().numpy()[:nnz], col)
    got = g.val.cpu().numpy()[:nnz]
    assert got.tobytes() == val.astype(np.float32).tobytes(), f"{name}: weights differ in bits"


@pytest.mark.parametrize("improved,loops,normalize", [(True, True, True), (False, False, True),
                                                       (False, False, False)])
def test_prep_options(ga, cref, improved, loops, normalize):
    n, ei = 300, random_multigraph(300, 2000, self_loops=40, dup=100, isolate=7)
    ew = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(1)) + 0.5
    g = ga.prepare_graph(ei.to(DEV), n, ew.to(DEV), add_self_loops=loops, improved=improved,
                         normalize=normalize)
    if normalize:
        s, d, w = cref.norm(ei.numpy(), ew.numpy(), n, add_self_loops=loops, fill=2.0 if improved else 1.0)
    else:
        s, d, w = ei[0].numpy(), ei[1].numpy(), ew.numpy()
    rowptr, col, val = csr_from_oracle(s, d, w, n)
    nnz = g.nnz()
    np.testing.assert_array_equal(g.rowptr.cpu().numpy(), rowptr)
    np.testing.assert_array_equal(g.col.cpu().numpy()[:nnz], col)
    assert g.val.cpu().numpy()[:nnz].tobytes() == val.astype(np.float32).tobytes()


def test_prep_rejects_bad_index(ga):
    ei = torch.tensor([[0, 1, 5], [1, 0, 2]])
    with pytest.raises(IndexError):
        ga.prepare_graph(ei.to(DEV), 3)
    with pytest.raises(TypeError):
        ga.prepare_graph(ei.to(DEV).int(), 6)
    with pytest.raises(ValueError):
        ga.prepare_graph(torch.zeros(3, 4, dtype=torch.long, device=DEV), 6)
    with pytest.raises(RuntimeError):
        ga.prepare_graph(ei, 6)


def test_transpose_matches_numpy(ga):
    n, ei = 300, random_multigraph(300, 2000, self_loops=40, dup=100, isolate=7)
    g = ga.prepare_graph(ei.to(DEV), n)
    nnz = g.nnz()
    rp, col, val = g.rowptr.cpu().numpy(), g.col.cpu().numpy()[:nnz], g.val.cpu().numpy()[:nnz]
    rows = np.repeat(np.arange(n), np.diff(rp))
    order = np.argsort(col, kind="stable")
    t_rp, t_col, t_val = (t.cpu().numpy() for t in g.transposed())
    np.testing.assert_array_equal(t_rp, np.concatenate([[0], np.cumsum(np.bincount(col, minlength=n))]))
    np.testing.assert_array_equal(t_col[:nnz], rows[order])
    assert t_val[:nnz].tobytes() == val[order].tobytes()


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_grouped_layout(ga, case):
    """gwen_gcn_group8: same entries per row, rows padded to whole groups of 8 with weight 0 and a
    column that is already in the row, null group behind the last row."""
    name, n, ei = case
    g = ga.prepare_graph(ei.to(DEV), n)
    rp, col, val = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy()
    grp_t, gcol_t, gval_t = g.grouped()
    gcol, gval = gcol_t.cpu().numpy(), gval_t.cpu().numpy()
    lens = np.diff(rp)
    if grp_t is None:                                   # uniform layout: every row exactly one group
        assert n > 0 and ((lens >= 1) & (lens <= 8)).all()
        grp = 8 * np.arange(n + 1)
    else:
        assert n == 0 or not ((lens >= 1) & (lens <= 8)).all()
        grp = grp_t.cpu().numpy()
    assert grp[0] == 0 and (np.diff(grp) % 8 == 0).all()
    for r in range(n):
        ln = rp[r + 1] - rp[r]
        gl = grp[r + 1] - grp[r]
        assert gl == (ln + 7) // 8 * 8
        np.testing.assert_array_equal(gcol[grp[r]:grp[r] + ln], col[rp[r]:rp[r + 1]])
        assert gval[grp[r]:grp[r] + ln].tobytes() == val[rp[r]:rp[r + 1]].tobytes()
        assert (gval[grp[r] + ln:grp[r + 1]] == 0).all()
        if ln:
            assert (gcol[grp[r] + ln:grp[r + 1]] == col[rp[r]]).all()
    assert (gcol[grp[n]:grp[n] + 8] == 0).all() and (gval[grp[n]:grp[n] + 8] == 0).all()


# ------------------------------------------------------------------------------------------------
# K2: fused propagate, bit-exact against the sequential C oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("F", [1, 2, 3, 4, 8, 12, 16, 24, 32, 64, 100, 128, 256, 260, 1000])
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[7], CASES[3]], ids=[IDS[0], IDS[1], IDS[7], IDS[3]])
def test_propagate_bit_exact(ga, cref, case, F):
    from gwen_amd import ops
    name, n, ei = case
    gen = torch.Generator().manual_seed(SEED + F)
    h = torch.randn(n, F, generator=gen)
    b = torch.randn(F, generator=gen) * 0.1
    g = ga.prepare_graph(ei.to(DEV), n)
    s, d, w = cref.norm(ei.numpy(), None, n)
    for bias, relu in ((None, False), (b, True)):
        got = ops.propagate(g, h.to(DEV), None if bias is None else bias.to(DEV), relu).cpu().numpy()
        ref = cref.propagate(s, d, w, h.numpy(), None if bias is None else bias.numpy(), relu)
        assert got.tobytes() == ref.tobytes(), f"{name} F={F} relu={relu}: not bit-identical"


def test_propagate_members_and_determinism(ga, cref):
    from gwen_amd import ops
    name, n, ei = CASES[0]
    h = torch.randn(3, n, 64, generator=torch.Generator().manual_seed(SEED))
    g = ga.prepare_graph(ei.to(DEV), n)
    s, d, w = cref.norm(ei.numpy(), None, n)
    a = ops.propagate(g, h.to(DEV)).cpu()
    b = ops.propagate(g, h.to(DEV)).cpu()
    assert torch.equal(a, b)
    for m in range(3):
        ref = cref.propagate(s, d, w, h[m].numpy())
        assert a[m].numpy().tobytes() == ref.tobytes()


# ------------------------------------------------------------------------------------------------
# K3: fp32-MFMA projection
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,fin,fout", [(1002, 8, 8), (1002, 64, 64), (777, 64, 32), (130, 16, 32),
                                            (1002, 256, 256), (125, 1000, 48), (33, 7, 5), (64, 3, 1),
                                            (10, 0, 4), (129, 100, 260), (125, 4096, 512), (40, 2048, 96)])
def test_linear(ga, cref, rows, fin, fout):
    from gwen_amd import ops
    gen = torch.Generator().manual_seed(SEED + rows + fin)
    x = torch.randn(rows, fin, generator=gen)
    w, b = make_params(fin, fout)
    got = ops.linear(x.to(DEV), w.to(DEV)).cpu()
    ref64 = x.double() @ w.double().t()
    assert rel_err(got, ref64) <= 1e-5
    chain = cref.linear(x.numpy(), w.numpy(), fma=True)           # k-ordered fmaf chain
    assert got.numpy().tobytes() == chain.tobytes(), "fp32 MFMA is not the k-ordered fmaf chain"
    got2 = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True).cpu()
    ref2 = torch.relu(torch.from_numpy(chain) + b)
    assert torch.equal(got2, ref2)
    # 3xbf16 split variant: fp32-class accuracy, not bitwise
    got3 = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True, exact=False).cpu()
    ref3 = torch.relu(ref64 + b.double())
    assert rel_err(got3, ref3) <= 2e-5
    # bf16x6 split (three images per operand, six terms): as close to fp64 as the exact fp32 MFMA, not bitwise
    got6 = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True, contract="bf16x6").cpu()
    assert rel_err(got6, ref3) <= 2e-6
    assert rel_err(got6, ref3) <= rel_err(got3, ref3) + 1e-7


@pytest.mark.parametrize("rows,fin,fout", [(16384, 256, 256), (20001, 64, 64), (40007, 128, 320), (100002, 256, 768),
                                            (16390, 64, 192), (33000, 128, 64)])
def test_linear_tall(ga, rows, fin, fout):
    """From 16 384 rows up the 3xbf16 projection runs on the K8 pipeline without a graph (wide.hip DENSE: rows
    DMA-staged through LDS, output columns in groups of 256 / 128 / 64).  Same split, same MFMA sequence as the
    128 x 128-tile kernel short inputs take: bitwise equal to it, and fp32-class against fp64."""
    from gwen_amd import ops, _lib
    from gwen_amd.graph import _ptr, _stream
    gen = torch.Generator().manual_seed(SEED + rows + fout)
    x = torch.randn(rows, fin, generator=gen).to(DEV)
    w, b = make_params(fin, fout)
    w, b = w.to(DEV), b.to(DEV)
    got = ops.linear(x, w, b, relu=True, exact=False)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert rel_err(got.cpu(), ref.cpu()) <= 2e-5
    for lo in (0, rows - 8000):                                   # 8 000 rows at a time: the short kernel
        part = ops.linear(x[lo:lo + 8000].contiguous(), w, b, relu=True, exact=False)
        assert torch.equal(part, got[lo:lo + 8000])
    plain = ops.linear(x, w, None, relu=False, exact=False)
    assert rel_err(plain.cpu(), (x.double() @ w.double().t()).cpu()) <= 2e-5
    # rows at a pitch (the C ABI's ldx / ldh): the left fin columns of a wider tensor into the right of another
    wide_x = torch.randn(rows, fin + 64, generator=gen).to(DEV)
    wide_o = torch.full((rows, fout + 32), -7.0, device=DEV)
    rc = _lib.lib().gwen_gcn_linear_f32(_ptr(wide_x), _ptr(w), _ptr(b), wide_o.data_ptr() + 32 * 4, rows, fin, fout,
                                        fin + 64, fout + 32, 1, 0, None, 0, _stream(torch.device(DEV)))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(wide_o[:, 32:], ops.linear(wide_x[:, :fin].contiguous(), w, b, relu=True, exact=False))
    assert bool((wide_o[:, :32] == -7.0).all())


# ------------------------------------------------------------------------------------------------
# one layer and the whole model against the torch oracle (the "reference PyTorch CPU path")
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("order", ["transform_first", "aggregate_first", "auto", "fused_exact"])
@pytest.mark.parametrize("fin,fout", [(8, 8), (64, 32), (16, 32), (64, 64), (256, 256), (6, 10),
                                      (128, 128), (16, 128), (128, 16), (32, 32)])
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[7]], ids=[IDS[0], IDS[1], IDS[7]])
def test_layer_vs_oracle(ga, cref, case, fin, fout, order):
    from oracle import gcn_oracle as O
    name, n, ei = case
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(fin, fout)
    from gwen_amd import ops
    if order == "fused_exact" and not ops.layer_supported(fin, fout):
        pytest.skip("K4 widths only")
    conv = ga.GCNConv(fin, fout).to(DEV)
    conv.order = order
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
        got = conv(x.to(DEV), ei.to(DEV)).cpu()
    ref = O.gcn_conv(x, ei, w, b)
    ref64 = torch.from_numpy(cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), f64=True))
    assert rel_err(got, ref) <= REL_TOL
    assert rel_err(got, ref64) <= REL_TOL
    # exact paths (explicit orders) should be no further from fp64 truth than the fp32 oracle is (x4 slack);
    # the default contraction of an AUTO layer (bf16x6, whatever kernels it resolves to) gets the exact paths'
    # slack: it carries 24 bits per operand
    slack = 1e-6
    assert rel_err(got, ref64) <= 4 * rel_err(ref, ref64) + slack
    if order == "auto":       # the faster tier on the same kernels: 2e-5
        conv.precision = "3xbf16"
        with torch.no_grad():
            got3 = conv(x.to(DEV), ei.to(DEV)).cpu()
        assert rel_err(got3, ref64) <= 4 * rel_err(ref, ref64) + 2e-5


@pytest.mark.parametrize("C,H,nu", [(8, 16, 10), (8, 8, 10), (64, 64, 10), (20, 48, 4)])
def test_model_vs_oracle(ga, C, H, nu):
    from oracle import gcn_oracle as O
    m = ga.geodesic_mesh(nu)
    ei = torch.from_numpy(m.edge_index)
    torch.manual_seed(SEED)
    ref_model = O.OracleGNNModel(O.OracleGNNConfig(m.num_nodes, m.num_nodes, C, C, H))
    with torch.no_grad():
        for p in ref_model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    model = ga.GNNModel(ga.GNNConfig(m.num_nodes, m.num_nodes, C, C, H))
    model.load_state_dict(ref_model.state_dict(), strict=True)
    model = model.to(DEV).eval()
    x = torch.randn(m.num_nodes, C)
    with torch.no_grad():
        got = model(x.to(DEV), ei.to(DEV)).cpu()
        ref = ref_model(x, ei)
        ref64 = ref_model.double()(x.double(), ei)
    assert got.shape == ref.shape
    assert rel_err(got, ref) <= REL_TOL
    assert rel_err(got, ref64) <= REL_TOL


def test_model_members_axis(ga):
    """[members, N, C] input == the same members run one by one (shared prepared graph)."""
    m = ga.geodesic_mesh(6)
    ei = torch.from_numpy(m.edge_index).to(DEV)
    torch.manual_seed(SEED)
    model = ga.GNNModel(ga.GNNConfig(1, 1, 16, 16, 32)).to(DEV).eval()
    x = torch.randn(4, m.num_nodes, 16, device=DEV)
    with torch.no_grad():
        batched = model(x, ei)
        single = torch.stack([model(x[i], ei) for i in range(4)])
    assert torch.equal(batched, single)


# ------------------------------------------------------------------------------------------------
# K4 fused layer and the whole-stack launcher
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fin,fout", [(16, 16), (16, 32), (32, 16), (64, 64), (64, 32), (32, 64),
                                      (128, 64), (64, 128), (128, 128), (16, 128), (256, 256), (256, 64),
                                      (32, 256)])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_fused_layer_vs_oracle(ga, cref, case, fin, fout):
    from gwen_amd import ops
    name, n, ei = case
    assert ops.layer_supported(fin, fout)
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(fin, fout)
    g = ga.prepare_graph(ei.to(DEV), n)
    ref64 = torch.from_numpy(cref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True))
    got = ops.layer_fused(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True).cpu()     # 3xbf16 contraction
    assert rel_err(got, ref64) <= 2e-5
    again = ops.layer_fused(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True).cpu()
    assert torch.equal(got, again)                               # atomic-free => reproducible
    exact = ops.layer_fused(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True, exact=True).cpu()
    assert rel_err(exact, ref64) <= 2e-6                         # fp32 MFMA contraction
    x6 = ops.layer_fused(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True, contract="bf16x6").cpu()
    assert rel_err(x6, ref64) <= 2e-6                            # bf16x6: fp32-class
    assert torch.equal(x6, ops.layer_fused(g, x.to(DEV), w.to(DEV), b.to(DEV), relu=True, contract="bf16x6").cpu())
    nob = ops.layer_fused(g, x.to(DEV), w.to(DEV)).cpu()
    ref_nob = torch.from_numpy(cref.conv(x.numpy(), ei.numpy(), w.numpy(), None, f64=True))
    assert rel_err(nob, ref_nob) <= 2e-5


def test_fused_layer_unsupported_widths_are_refused(ga):
    from gwen_amd import ops, _lib
    assert not ops.layer_supported(8, 8) and not ops.layer_supported(512, 512)
    assert not ops.layer_supported(24, 64)
    g = ga.prepare_graph(CASES[5][2].to(DEV), 3)
    with pytest.raises(_lib.GwenHipError):
        ops.layer_fused(g, torch.zeros(3, 24, device=DEV), torch.zeros(64, 24, device=DEV))


@pytest.mark.parametrize("pre", [False, True])
@pytest.mark.parametrize("fin,f1,f2", [(64, 64, 32), (32, 32, 16), (16, 64, 16), (64, 128, 64), (64, 128, 32),
                                       (32, 64, 16)])
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[7]], ids=[IDS[0], IDS[1], IDS[7]])
def test_chained_kernel_vs_oracle(ga, cref, case, fin, f1, f2, pre):
    """K5 against the fp64 C oracle: layer l, then layer l+1's projection (pre=False); or bias/ReLU of
    a pre-projected layer, then the next projection (pre=True)."""
    from gwen_amd import ops
    name, n, ei = case
    g = ga.prepare_graph(ei.to(DEV), n)
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(SEED))
    w1, b1 = make_params(fin, f1)
    w2, _ = make_params(f1, f2)
    if pre:
        bpre = torch.randn(fin, generator=torch.Generator().manual_seed(SEED + 2)) * 0.1
        agg = cref.conv(x.numpy(), ei.numpy(), np.eye(fin, dtype=np.float32), bpre.numpy(), relu=True, f64=True)
        want = agg @ w1.double().numpy().T
        got = ops.chain(g, x.to(DEV), w1.to(DEV), None, bpre.to(DEV), relu=True, pre=True).cpu()
    else:
        y1 = cref.conv(x.numpy(), ei.numpy(), w1.numpy(), b1.numpy(), relu=True, f64=True)
        want = y1 @ w2.double().numpy().T
        got = ops.chain(g, x.to(DEV), w1.to(DEV), w2.to(DEV), b1.to(DEV), relu=True, pre=False).cpu()
    assert got.shape == want.shape
    assert rel_err(got, want) <= 3e-5


@pytest.mark.parametrize("F", [16, 32, 64, 128])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_grouped_gather_kernel(ga, cref, case, F):
    """K5 with nothing chained (activation-first propagate on the grouped layout) == K2 to rounding."""
    from gwen_amd import ops, _lib
    from gwen_amd.graph import _ptr, _stream
    name, n, ei = case
    g = ga.prepare_graph(ei.to(DEV), n)
    h = torch.randn(n, F, generator=torch.Generator().manual_seed(SEED)).to(DEV)
    b = (torch.randn(F, generator=torch.Generator().manual_seed(SEED + 1)) * 0.1).to(DEV)
    want = ops.propagate(g, h, b, True)
    out = torch.empty_like(h)
    gr, gc, gv = g.grouped()
    rc = _lib.lib().gwen_gcn_chain_f32(None if gr is None else _ptr(gr), _ptr(gc), _ptr(gv), _ptr(h), None, None, _ptr(b), _ptr(out),
                                       n, F, 0, 0, 1, 1, 1, n * F, n * F, _lib.CONTRACT_BF16X3, _stream(h.device))
    assert rc == 0
    assert rel_err(out, want) <= 2e-6


@pytest.mark.parametrize("members", [1, 3])
@pytest.mark.parametrize("C,H", [(64, 64), (8, 16), (20, 48), (16, 256), (128, 128), (32, 64)])
def test_stack_forward_equals_layer_by_layer(ga, members, C, H):
    """gwen_gnn_forward_f32 (one host call) == the per-layer autograd path, bit for bit."""
    m = ga.geodesic_mesh(7)
    ei = torch.from_numpy(m.edge_index).to(DEV)
    torch.manual_seed(SEED)
    model = ga.GNNModel(ga.GNNConfig(1, 1, C, C, H)).to(DEV)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    x = torch.randn(members, m.num_nodes, C, device=DEV)
    x = x[0] if members == 1 else x
    g = model.prepare(ei, m.num_nodes)
    with torch.no_grad():
        one_call = model(x, g)
    layered = model.conv_layers(x.clone().requires_grad_(), g).detach()
    # the launcher re-brackets shrinking layers (K5 chains their projection), so equality is to
    # rounding, not to the bit; with explicit per-layer orders it is bit for bit
    assert rel_err(one_call, layered) <= 2e-5
    # with the same explicit per-layer orders on both sides it is bit for bit
    convs = [c for c in model.modules() if isinstance(c, ga.GCNConv)]
    for c in convs:
        c.order = "fused" if ga.ops.layer_supported(c.in_channels, c.out_channels) else \
            ("aggregate_first" if c.in_channels < c.out_channels else "transform_first")
    layered_explicit = model.conv_layers(x.clone().requires_grad_(), g).detach()
    assert torch.equal(ga.StackForward(model.stack(), g).run(x), layered_explicit)
    assert rel_err(layered_explicit, layered) <= 2e-5
    for c in convs:
        c.order = "auto"
    ev = ga.KernelEvents(12)
    again = ga.StackForward(model.stack(), g).run(x, events=ev)
    assert torch.equal(again, one_call)
    d = ev.durations()
    assert 6 <= len(d) <= 12 and all(t >= 0 for *_, t in d)
    graphed = ga.GraphedForward(ga.StackForward(model.stack(), g), x)      # hipGraph replay
    assert torch.equal(graphed(), one_call)
    x2 = torch.randn_like(x)
    with torch.no_grad():
        assert torch.equal(graphed(x2), model(x2, g))


# ------------------------------------------------------------------------------------------------
# known-answer tests on the device
# ------------------------------------------------------------------------------------------------
def test_kat_complete_graph_is_mean(ga):
    """On K_N (the reference's own graph, utils.py:176) every output row is mean_j(x_j W^T) + b."""
    n, fin, fout = 125, 32, 16
    ei = torch.from_numpy(ga.complete_graph(n)).to(DEV)
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(fin, fout)
    conv = ga.GCNConv(fin, fout).to(DEV)
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
        got = conv(x.to(DEV), ei).cpu().double()
    want = (x.double() @ w.double().t()).mean(0, keepdim=True) + b.double()
    assert rel_err(got, want.expand_as(got)) <= 1e-5


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-6), ("f16x3", 1e-6), ("bf16x6", 1e-6), ("3xbf16", 2e-5)])
def test_kat_path_and_cycle(ga, precision, tol):
    fin = fout = 4
    eye = torch.eye(4)
    conv = ga.GCNConv(fin, fout).to(DEV)
    conv.precision = precision
    with torch.no_grad():
        conv.lin.weight.copy_(eye); conv.bias.zero_()
    x = torch.randn(3, 4, generator=torch.Generator().manual_seed(SEED)).double()
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    got = conv(x.float().to(DEV), ei.to(DEV)).detach().cpu().double()
    r6 = 1 / 6 ** 0.5
    want = torch.stack([x[0] / 2 + x[1] * r6, x[0] * r6 + x[1] / 3 + x[2] * r6, x[1] * r6 + x[2] / 2])
    assert rel_err(got, want) <= tol
    x = torch.randn(6, 4, generator=torch.Generator().manual_seed(SEED)).double()
    ei = torch.tensor([[0, 1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 0]])
    got = conv(x.float().to(DEV), ei.to(DEV)).detach().cpu().double()
    want = 0.5 * (x + torch.roll(x, 1, 0))
    assert rel_err(got, want) <= tol


def test_properties_full_size(ga):
    """BASELINE config-2 size (N=100 002, E=600 000, F=64): oracle comparison + linearity +
    permutation equivariance + run-to-run determinism of the whole layer."""
    from oracle import gcn_oracle as O
    m = ga.geodesic_mesh(100)
    n = m.num_nodes
    ei = torch.from_numpy(m.edge_index)
    gen = torch.Generator().manual_seed(SEED)
    x = torch.randn(n, 64, generator=gen)
    y = torch.randn(n, 64, generator=gen)
    w, b = make_params(64, 64)
    conv = ga.GCNConv(64, 64).to(DEV)
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
        eid = ei.to(DEV)
        fx = conv(x.to(DEV), eid)
        assert torch.equal(fx, conv(x.to(DEV), eid))
        ref = O.gcn_conv(x, ei, w, b)
        assert rel_err(fx, ref) <= REL_TOL
        fy = conv(y.to(DEV), eid)
        fxy = conv((2 * x - y).to(DEV), eid)
        lin = 2 * (fx - b.to(DEV)) - (fy - b.to(DEV)) + b.to(DEV)
        assert rel_err(fxy, lin) <= 1e-5
        perm = torch.randperm(n, generator=gen)
        inv = torch.empty_like(perm); inv[perm] = torch.arange(n)
        fp = conv(x[perm].to(DEV), inv[ei].to(DEV))          # node i -> position inv[i]
        assert rel_err(fp, fx[perm.to(DEV)]) <= 1e-5


# ------------------------------------------------------------------------------------------------
# backward: gradients against torch autograd through the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("order,fin,fout", [("transform_first", 24, 40), ("aggregate_first", 24, 40),
                                            ("fused", 32, 64), ("fused", 64, 16), ("fused_exact", 16, 128)])
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[7]], ids=[IDS[0], IDS[1], IDS[7]])
def test_layer_backward(ga, case, order, fin, fout):
    from oracle import gcn_oracle as O
    name, n, ei = case
    x = torch.randn(n, fin, generator=torch.Generator().manual_seed(SEED))
    w, b = make_params(fin, fout)
    gout = torch.randn(n, fout, generator=torch.Generator().manual_seed(SEED + 1))
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    torch.relu(O.gcn_conv(xr, ei, wr, br)).backward(gout)
    conv = ga.GCNConv(fin, fout).to(DEV)
    conv.order = order
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
    xd = x.to(DEV).requires_grad_()
    conv(xd, ei.to(DEV), relu=True).backward(gout.to(DEV))
    assert rel_err(xd.grad, xr.grad) <= REL_TOL
    assert rel_err(conv.lin.weight.grad, wr.grad) <= REL_TOL
    assert rel_err(conv.bias.grad, br.grad) <= REL_TOL


def test_model_training_step(ga):
    """One optimiser step of the reference's loop shape (models_gnn.py:362-373) moves the loss."""
    m = ga.geodesic_mesh(4)
    ei = torch.from_numpy(m.edge_index).to(DEV)
    torch.manual_seed(SEED)
    model = ga.GNNModel(ga.GNNConfig(1, 1, 8, 8, 16)).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    x = torch.randn(m.num_nodes, 8, device=DEV)
    mask = torch.zeros(m.num_nodes, dtype=torch.bool, device=DEV); mask[::3] = True
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = ga.loss_func(model(x, ei), x, mask)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    live = (".conv1.", ".conv2.", ".conv3.", ".upconv3.", ".upconv4.", ".upconv5.")
    for n_, p in model.named_parameters():
        assert (p.grad is not None) == any(k in n_ for k in live), n_


# ------------------------------------------------------------------------------------------------
# whole-stack training path: gwen_gnn_forward_f32 (acts) + gwen_gnn_backward_f32, one host call each
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,GRAD_TOL", [("f16x3", 1e-5), ("3xbf16", 1e-4)])
@pytest.mark.parametrize("members", [1, 3])
@pytest.mark.parametrize("C,H", [(64, 64), (8, 16), (20, 48), (32, 128), (16, 256)])
def test_stack_backward_vs_oracle_autograd(ga, C, H, members, precision, GRAD_TOL):
    """Every gradient of GNNModel (x, 6 weights, 6 biases) through the stack launchers against torch autograd
    on the CPU oracle -- widths K4's backward kernel takes (fused launch) and widths it does not (K2^T + K3).
    The backward contracts on the layer's own precision: 1e-5 on the fp32-class default (the reference's fp32
    autograd through loss.backward(), /root/reference/src/gwen/models_gnn.py:372), 1e-4 on the "3xbf16" tier."""
    from oracle import gcn_oracle as O
    m = ga.geodesic_mesh(7, reorder="hilbert")
    n = m.num_nodes                                     # 492 > 256: not the K7 path
    ei = torch.from_numpy(m.edge_index)
    torch.manual_seed(SEED)
    ref = O.OracleGNNModel(O.OracleGNNConfig(n, n, C, C, H))
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    model = ga.GNNModel(ga.GNNConfig(n, n, C, C, H))
    model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(DEV).set_precision(precision)
    x = torch.randn(members, n, C, generator=torch.Generator().manual_seed(SEED))
    gout = torch.randn(members, n, C, generator=torch.Generator().manual_seed(SEED + 1))
    xr = x.clone().requires_grad_()
    torch.stack([ref(xr[k], ei) for k in range(members)]).backward(gout)
    xd = (x if members > 1 else x[0]).to(DEV).requires_grad_()
    out = model(xd, ei.to(DEV))
    assert out.grad_fn is not None and "GNNStackFunction" in type(out.grad_fn).__name__
    saved = out.grad_fn.saved_tensors                     # (x, act_1 .. act_6): read before backward frees them
    assert len(saved) == 7
    masks = [(a > 0).view(members, n, -1).cpu() for a in saved[1:6]]      # the device's five ReLU patterns
    out.backward((gout if members > 1 else gout[0]).to(DEV))
    with torch.no_grad():
        want = torch.stack([ref(x[k], ei) for k in range(members)])
    assert rel_err(out.detach().view(members, n, C), want) <= REL_TOL

    def l2_err(a, b):
        a, b = a.double().cpu(), b.double()
        return float((a - b).norm() / b.norm())

    # PRIMARY gate (1e-4 on every member and every parameter): the oracle's backward at the DEVICE's activation
    # patterns.  A pre-activation within rounding of zero can fall on the other side of a ReLU than in the
    # oracle's own forward; the device's patterns are read back (acts > 0 of the saved layer outputs) and the
    # oracle is differentiated at exactly those (O.forward_with_masks, fp64), so a flipped unit can not explain
    # any difference and the comparison is kernel arithmetic only.
    import copy
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad()
    xm = x.double().requires_grad_()
    torch.stack([O.forward_with_masks(ref64, xm[k], ei, [mk[k] for mk in masks]) for k in range(members)]) \
        .backward(gout.double())
    got = dict(model.named_parameters())
    for k in range(members):
        assert l2_err(xd.grad.view(members, n, C)[k], xm.grad[k]) <= GRAD_TOL, (k, "grad_x")
        assert rel_err(xd.grad.view(members, n, C)[k], xm.grad[k]) <= GRAD_TOL, (k, "grad_x max")
    for name, p in ref64.named_parameters():
        if p.grad is None:
            assert got[name].grad is None, name
        else:
            assert l2_err(got[name].grad, p.grad) <= GRAD_TOL, name
            assert rel_err(got[name].grad, p.grad) <= GRAD_TOL, name
    # SECONDARY sanity bound against the oracle's OWN patterns (what the reference's training would compute):
    # flipped units move a member's gradient by 1e-4..1e-2 in their neighbourhood, never more
    errs = sorted(l2_err(xd.grad.view(members, n, C)[k], xr.grad[k]) for k in range(members))
    assert errs[len(errs) // 2] <= 1e-3 and errs[-1] <= 5e-2, errs
    for name, p in ref.named_parameters():
        if p.grad is not None:
            assert l2_err(got[name].grad, p.grad) <= 2e-2, name
    # the per-layer autograd path computes the same gradients
    model.zero_grad()
    xd2 = (x if members > 1 else x[0]).to(DEV).requires_grad_()
    stack_grads = {k: v.grad.clone() for k, v in got.items() if v.grad is not None}
    model.zero_grad()
    model.conv_layers(xd2, model.prepare(ei.to(DEV), n)).backward((gout if members > 1 else gout[0]).to(DEV))
    assert rel_err(xd2.grad, xd.grad) <= 2e-5          # same forward values, same masks: rounding order only
    for k, v in stack_grads.items():
        assert rel_err(got[k].grad, v) <= 2e-5, k


@pytest.mark.parametrize("contract,tol", [("bf16x6", 2e-6), ("f16x3", 2e-6), ("3xbf16", 2e-5), ("fp32", 2e-6)])
@pytest.mark.parametrize("rows,fin,fout", [((700,), 64, 64), ((3, 211), 6, 64), ((1000,), 256, 10), ((5, 40), 30, 7)])
@pytest.mark.parametrize("relu,use_bias", [(False, True), (True, True), (True, False)])
def test_linear_autograd_vs_fp64(ga, contract, tol, rows, fin, fout, relu, use_bias):
    """ops.linear_autograd (K3 with the hand-written backward: g_x = g W, g_W = g^T x, g_b = column sums, ReLU mask on
    y) against fp64 torch autograd -- 2-D and 3-D inputs, widths the vector K3 path does not take (Fin = 6, 30), with
    and without bias / ReLU.  (The forecaster's embedding and read-out train through it.)"""
    from gwen_amd import ops
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(*rows, fin, generator=g)
    w, b = make_params(fin, fout)
    gout = torch.randn(*rows, fout, generator=g)
    xd = x.to(DEV).requires_grad_()
    wd = w.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if use_bias else None
    y = ops.linear_autograd(xd, wd, bd, relu=relu, contract=contract)
    y.backward(gout.to(DEV))
    x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
    b64 = b.double().requires_grad_() if use_bias else None
    pre = torch.nn.functional.linear(x64, w64, b64)
    # differentiate the fp64 expression at the DEVICE's ReLU pattern (a unit within rounding of zero may fall either side)
    y64 = pre * (y.detach().cpu() > 0).double() if relu else pre
    y64.backward(gout.double())
    assert rel_err(y.detach(), y64.detach()) <= tol
    assert rel_err(xd.grad, x64.grad) <= tol
    assert rel_err(wd.grad, w64.grad) <= tol
    if use_bias:
        assert rel_err(bd.grad, b64.grad) <= tol


@pytest.mark.parametrize("rows,fout,fin", [(5000, 128, 128), (70001, 256, 256), (3000, 256, 64), (2047, 64, 256),
                                           (2049, 128, 256), (1, 256, 256), (4096, 192, 128), (300, 64, 64), (777, 20, 48),
                                           (200000, 64, 64), (150001, 128, 64), (100002, 32, 16), (5000, 16, 32),
                                           (1, 16, 16), (513, 64, 32), (40000, 48, 64)])
def test_grad_weight_every_contraction_vs_fp64(ga, rows, fout, fin):
    """ops.grad_weight = g^T x on every contraction against fp64: widths the split kernels take (multiples of 64 from
    64 x 64: 64 x 64 tiles, LDS-staged, 256 .. 2 048 rows per block by row count -- one row, one row short of / past a
    chunk, a 192-wide side, row counts whose partial sums fill the workspace to its last slot) and widths that stay on
    the fp32-input MFMA whatever is asked (one / two 32 x 32 tiles: their row steps dealt over the block's waves; more: a wave a
    tile); rows of very different magnitude; two runs
    bitwise equal (fixed summation order)."""
    from gwen_amd import ops
    gen = torch.Generator().manual_seed(SEED + rows)
    g = torch.randn(rows, fout, generator=gen) * torch.exp2(torch.randint(-6, 7, (rows, 1), generator=gen).float())
    x = torch.randn(rows, fin, generator=gen)
    want = g.double().t() @ x.double()
    gd, xd = g.to(DEV), x.to(DEV)
    for contract, tol in ((None, 2e-6), ("fp32", 2e-6), ("bf16x6", 4e-6), ("f16x3", 4e-6), ("3xbf16", 3e-5)):
        got = ops.grad_weight(gd, xd, contract)
        assert rel_err(got, want) <= tol, (contract, rel_err(got, want))
        assert torch.equal(got, ops.grad_weight(gd, xd, contract)), contract


def test_grad_batch_many_reductions_one_finish(ga):
    """ops.GradBatch: 40 weight / bias reductions of mixed shape (more than one 32-task launch; one-chunk reductions
    that finish in stage 1; an empty one) -- each against fp64, and bitwise the stand-alone ops.grad_weight /
    ops.grad_bias of the same operands in spirit: the same chunking, so equal to 1e-6 and run to run bitwise."""
    from gwen_amd import ops
    gen = torch.Generator().manual_seed(SEED)
    shapes = [(70001, 256, 256), (5000, 128, 128), (100, 64, 64), (2049, 128, 256), (777, 20, 48), (0, 64, 64),
              (200000, 64, 64), (1, 256, 64)] * 3
    ops_in = [(torch.randn(r, fo, generator=gen).to(DEV), torch.randn(r, fi, generator=gen).to(DEV))
              for r, fo, fi in shapes[:8]] * 3
    runs = []
    for _ in range(2):
        gb = ops.GradBatch()
        outs = []
        for k, (g, x) in enumerate(ops_in):
            outs.append(gb.grad_weight(g, x, ("bf16x6", "f16x3", None)[k % 3]))
            if k % 2 == 0 or k >= 8:
                outs.append(gb.grad_bias(g))
        gb.finish()
        torch.cuda.synchronize()
        runs.append(outs)
    assert len(runs[0]) >= 40 and all(torch.equal(a, b) for a, b in zip(*runs))
    it = iter(runs[0])
    for k, (g, x) in enumerate(ops_in):
        got = next(it)
        assert rel_err(got, g.double().t() @ x.double()) <= 4e-6 if g.size(0) else not bool(got.any())
        if g.size(0):
            assert rel_err(got, ops.grad_weight(g, x, ("bf16x6", "f16x3", None)[k % 3])) <= 1e-6
        if k % 2 == 0 or k >= 8:
            gotb = next(it)
            assert rel_err(gotb, g.double().sum(0)) <= 2e-6 if g.size(0) else not bool(gotb.any())


@pytest.mark.parametrize("rows,fout,fin", [(70001, 256, 256), (5000, 128, 128), (600, 64, 64), (2049, 128, 256),
                                           (100002, 64, 128), (3000, 192, 64), (777, 20, 48), (1, 64, 64)])
def test_grad_weight_bias_one_launch(ga, rows, fout, fin):
    """GradBatch.grad_weight_bias: grad_W AND the column sums of the same g from one stage-1 launch where the LDS-staged
    kernel takes the weight gradient (every block shape: 128 x 128, 128 x 64, 64 x 128, 64 x 64; one chunk and many),
    the two separate launches elsewhere (20 x 48) -- both against fp64, the weight gradient bitwise the plain one."""
    from gwen_amd import ops
    gen = torch.Generator().manual_seed(SEED + rows + fout)
    g = (torch.randn(rows, fout, generator=gen) * torch.exp2(torch.randint(-6, 7, (rows, 1), generator=gen).float())).to(DEV)
    x = torch.randn(rows, fin, generator=gen).to(DEV)
    for contract in ("3xbf16", "bf16x6"):
        runs = []
        for _ in range(2):
            gb = ops.GradBatch()
            w, b = gb.grad_weight_bias(g, x, contract)
            gb.finish()
            runs.append((w, b))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
        w, b = runs[0]
        assert torch.equal(w, ops.grad_weight(g, x, contract)) or rel_err(w, ops.grad_weight(g, x, contract)) <= 1e-6
        assert rel_err(w, g.double().t() @ x.double()) <= (4e-6 if contract == "bf16x6" else 3e-5)
        assert rel_err(b, g.double().sum(0)) <= 2e-6, rel_err(b, g.double().sum(0))


@pytest.mark.parametrize("fg,fx,members", [(64, 64, 1), (32, 16, 3), (16, 32, 2), (64, 32, 1), (128, 64, 1), (256, 256, 1)])
def test_layer_bwd_leaves_the_bias_sums_of_the_layer_below(ga, fg, fx, members):
    """gwen_gcn_layer_bwd_bias_f32 through the C ABI: the same gh / gx as gwen_gcn_layer_bwd_f32 bit for bit, and -- narrow
    layers -- one partial row per (member, chunk of rows) whose sum is the column sum of the masked gx (stage 1 of the grad_b
    of the layer below); the persistent wide kernels report 0 chunks and the caller keeps the separate reduction."""
    import ctypes as C
    from gwen_amd import _lib
    from gwen_amd.graph import _ptr, _stream
    m = ga.geodesic_mesh(9, reorder="hilbert")
    n = m.num_nodes
    g = ga.prepare_graph(torch.from_numpy(m.edge_index).to(DEV), n)
    tr, tc, tv = g.transposed_grouped()
    gen = torch.Generator().manual_seed(SEED + fg + fx)
    grad = torch.randn(members, n, fg, generator=gen).to(DEV)
    wt = (torch.randn(fx, fg, generator=gen) / fg ** 0.5).to(DEV)
    mask = torch.randn(members, n, fx, generator=gen).to(DEV)
    L = _lib.lib()
    dev = torch.device(DEV)
    outs = []
    for with_bias in (False, True):
        gh, gx = torch.empty_like(grad), torch.empty(members, n, fx, device=DEV)
        rows = int(L.gwen_gcn_layer_bwd_bias_rows(n, members))
        part = torch.full((rows, fx), float("nan"), device=DEV)
        chunks = C.c_int64(-1)
        args = [_ptr(tr), _ptr(tc), _ptr(tv), _ptr(grad), _ptr(wt), _ptr(mask), _ptr(gh), _ptr(gx), n, fg, fx, members,
                _lib.CONTRACT_BF16X6]
        if with_bias:
            rc = L.gwen_gcn_layer_bwd_bias_f32(*args, _ptr(part), C.cast(C.pointer(chunks), C.c_void_p), _stream(dev))
        else:
            rc = L.gwen_gcn_layer_bwd_f32(*args, _stream(dev))
        _lib.check(rc, "layer_bwd")
        torch.cuda.synchronize()
        outs.append((gh, gx, part, chunks.value))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    gh, gx, part, nch = outs[1]
    assert bool((gx[mask <= 0] == 0).all())
    if fg * fx >= 128 * 128:
        assert nch == 0
    else:
        assert 0 < nch <= part.size(0) and bool(torch.isfinite(part[:nch]).all()) and bool(torch.isnan(part[nch:]).all())
        assert rel_err(part[:nch].double().sum(0), gx.double().sum((0, 1))) <= 2e-6


@pytest.mark.parametrize("rows,k,n", [(125, 1024, 16384), (125, 16384, 1024), (300, 20, 48), (1000, 64, 64), (7, 125, 512),
                                      (129, 256, 100), (1, 32, 16), (513, 96, 130)])
def test_linear_nn_weight_as_stored(ga, rows, k, n):
    """ops.linear_nn = x [rows, K] @ Wt [K, N] with Wt row-major -- the backward's g W without a transposing copy
    (gwen_gcn_linear_nn_f32): the reference's own shapes (split-K over 16 384), widths that are no multiple of 4 or of a
    tile, one row; both splits against fp64, and bitwise the transposed form's result is NOT required (another slab
    order) but the same bar is."""
    from gwen_amd import ops
    gen = torch.Generator().manual_seed(SEED + rows + k)
    x = torch.randn(rows, k, generator=gen).to(DEV)
    wt = (torch.randn(k, n, generator=gen) / k ** 0.5).to(DEV)
    want = x.double() @ wt.double()
    # ("fp32": the transposed exact kernel, a k-ordered fp32 fmaf chain -- its rounding grows with the 16 384-long sum)
    for contract, tol in (("bf16x6", 2e-6), ("f16x3", 2e-6), ("3xbf16", 3e-5), ("fp32", 2e-6 if k <= 1024 else 1e-5)):
        got = ops.linear_nn(x, wt, contract)
        assert got.shape == (rows, n)
        assert rel_err(got, want) <= tol, (contract, rel_err(got, want))
        assert torch.equal(got, ops.linear_nn(x, wt, contract))
        assert rel_err(got, ops.linear(x, wt.t().contiguous(), contract=contract)) <= tol


def test_stack_backward_without_input_grad_and_determinism(ga):
    m = ga.geodesic_mesh(9, reorder="hilbert")
    ei = torch.from_numpy(m.edge_index).to(DEV)
    torch.manual_seed(SEED)
    model = ga.GNNModel(ga.GNNConfig(1, 1, 64, 64, 64)).to(DEV)
    x = torch.randn(m.num_nodes, 64, device=DEV)                       # no grad wrt the input: as in training
    grads = []
    for _ in range(2):
        model.zero_grad()
        model(x, ei).square().mean().backward()
        grads.append([p.grad.clone() for p in model.parameters() if p.grad is not None])
    assert len(grads[0]) == 12 and all(torch.equal(a, b) for a, b in zip(*grads))


def test_precision_switch_train_and_eval_agree(ga):
    """One documented precision rule for both host paths: the per-layer autograd path (train) and the stack
    launcher (eval) give the same forward values -- bitwise for widths without K5 chains -- in both
    precisions, on K4 widths and on the K3 + K2 fallbacks; "fp32" is closer to the fp64 oracle than "3xbf16"."""
    from oracle import gcn_oracle as O
    m = ga.geodesic_mesh(7, reorder="hilbert")
    n, ei = m.num_nodes, torch.from_numpy(m.edge_index)
    for C, H in ((20, 48), (16, 64)):
        torch.manual_seed(SEED)
        ref = O.OracleGNNModel(O.OracleGNNConfig(n, n, C, C, H)).double()
        model = ga.GNNModel(ga.GNNConfig(n, n, C, C, H))
        model.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        model = model.to(DEV)
        x = torch.randn(n, C, generator=torch.Generator().manual_seed(SEED))
        with torch.no_grad():
            want = ref(x.double(), ei)
        errs = {}
        for prec in ("f16x3", "bf16x6", "3xbf16", "fp32"):
            model.set_precision(prec)
            assert all(c.precision == prec for c in model.modules() if isinstance(c, ga.GCNConv))
            with torch.no_grad():
                ev = model(x.to(DEV), ei.to(DEV))                       # stack launcher
            tr = model.conv_layers(x.to(DEV).requires_grad_(), model.prepare(ei.to(DEV), n)).detach()   # per layer
            if (C, H) == (20, 48) or prec == "fp32":                    # no K5 chain re-bracketing: bit for bit
                assert torch.equal(ev, tr), (C, H, prec)
            else:
                assert rel_err(ev, tr) <= 2e-5
            errs[prec] = rel_err(ev, want)
            assert errs[prec] <= REL_TOL
        assert errs["fp32"] <= 2e-6 and errs["fp32"] <= errs["3xbf16"]
        assert errs["bf16x6"] <= 2e-6 and errs["bf16x6"] <= errs["3xbf16"]
        assert errs["f16x3"] <= 2e-6 and errs["f16x3"] <= errs["3xbf16"]          # the default is fp32-class
    assert ga.GCNConv(8, 8).precision == "f16x3"
    with pytest.raises(ValueError):
        model.set_precision("bf16")


@pytest.mark.parametrize("members,n,c", [(1, 1002, 64), (3, 777, 8), (1, 100002, 64), (2, 5, 4)])
def test_masked_l1_fused_vs_reference_formula(ga, members, n, c):
    """loss_func on the GPU (gwen_masked_l1_f32: value + gradient in one pass) against the reference's own
    expression l1_loss(output[mask], target[mask]) (models_gnn.py:261-265) and its autograd, in fp64."""
    g = torch.Generator().manual_seed(SEED + n + c)
    shape = (n, c) if members == 1 else (members, n, c)
    o = torch.randn(*shape, generator=g)
    t = torch.randn(*shape, generator=g)
    t.view(-1)[::7] = o.view(-1)[::7]                                # exact zeros of the difference: sign(0) = 0
    mask = torch.rand(n, generator=g) < 0.5
    od = o.to(DEV).requires_grad_(True)
    loss = ga.loss_func(od, t.to(DEV), mask.to(DEV))
    loss.backward()
    o64 = o.double().requires_grad_(True)
    sel = (o64[mask], t.double()[mask]) if members == 1 else (o64[:, mask], t.double()[:, mask])
    want = torch.nn.functional.l1_loss(*sel)
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-6 * abs(float(want))
    assert torch.allclose(od.grad.cpu().double(), o64.grad, rtol=1e-6, atol=0)
    again = ga.loss_func(od.detach(), t.to(DEV), mask.to(DEV))       # fixed-order sums: bitwise reproducible
    assert torch.equal(again, loss.detach())
    half = ga.loss_func(od, t.to(DEV), mask.to(DEV)) * 0.5           # an upstream factor reaches the gradient
    od.grad = None
    half.backward()
    assert torch.allclose(od.grad.cpu().double(), 0.5 * o64.grad, rtol=1e-6, atol=0)


@pytest.mark.parametrize("offset,n", [(1, 1000), (7, 33), (15, 16), (3, 5), (0, 17), (9, 100002)])
def test_masked_l1_mask_at_any_byte_offset(ga, offset, n):
    """The picked-row count reads the mask 16 bytes at a time: masks that start at any byte offset of their storage (a view)
    and lengths with heads / tails shorter than a vector give the same loss as an aligned copy, and the reference formula."""
    g = torch.Generator().manual_seed(SEED + offset + n)
    o, t = torch.randn(n, 8, generator=g), torch.randn(n, 8, generator=g)
    store = torch.zeros(n + 32, dtype=torch.bool, device=DEV)
    mask = torch.rand(n, generator=g) < 0.4
    mask[n - 1] = True
    store[offset:offset + n] = mask.to(DEV)
    store[:offset] = True                                              # bytes outside the view must not be counted
    store[offset + n:] = True
    view = store[offset:offset + n]
    assert view.data_ptr() % 16 == offset % 16
    got = ga.loss_func(o.to(DEV), t.to(DEV), view)
    assert torch.equal(got, ga.loss_func(o.to(DEV), t.to(DEV), mask.to(DEV)))
    want = torch.nn.functional.l1_loss(o.double()[mask], t.double()[mask])
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))


def test_masked_l1_empty_mask_is_nan_like_torch(ga):
    o = torch.randn(64, 8, device=DEV)
    none = torch.zeros(64, dtype=torch.bool, device=DEV)
    assert torch.isnan(ga.loss_func(o, torch.zeros_like(o), none))


def test_training_step_replays_from_a_hipgraph(ga):
    """Forward + masked-L1 loss + backward + Adam of the reference's loop shape (models_gnn.py:362-373) captured ONCE
    into a hipGraph and replayed: every launcher of libgwen_hip.so is capturable (no allocation, no synchronisation
    inside), so a training step can leave the host's per-launch cost behind (tools/train_bench.py ... graph)."""
    m = ga.geodesic_mesh(9, reorder="hilbert")
    ei = torch.from_numpy(m.edge_index).to(DEV)
    torch.manual_seed(SEED)
    model = ga.GNNModel(ga.GNNConfig(1, 1, 16, 16, 32)).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True, capturable=True)
    x = torch.randn(m.num_nodes, 16, device=DEV)
    mask = torch.zeros(m.num_nodes, dtype=torch.bool, device=DEV); mask[::2] = True

    def step():
        opt.zero_grad(set_to_none=True)
        loss = ga.loss_func(model(x, ei), x, mask)
        loss.backward()
        opt.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss = ga.loss_func(model(x, ei), x, mask)
        loss.backward()
        opt.step()
    g.replay()
    torch.cuda.synchronize()
    first = float(loss.detach())
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    assert float(loss.detach()) < first


@pytest.mark.parametrize("contract", ["bf16x6", "3xbf16"])
def test_split_contractions_are_scale_free(ga, contract):
    """bf16 keeps fp32's exponent range, so the split contractions need no scaling: multiplying the input by a power
    of two (far from overflow / underflow) scales every image, every product and every partial sum by exactly that
    power -- the output is bitwise the scaled output.  (An fp16 split would not have this property.)"""
    from gwen_amd import ops
    m = ga.geodesic_mesh(6, reorder="hilbert")
    n = m.num_nodes
    g = ga.prepare_graph(torch.from_numpy(m.edge_index).to(DEV), n)
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(SEED)).to(DEV)
    w, _ = make_params(64, 64)
    w = w.to(DEV)
    base = ops.layer_fused(g, x, w, None, relu=True, contract=contract)
    for p in (40, -40, 90, -90):
        s = 2.0 ** p
        got = ops.layer_fused(g, x * s, w, None, relu=True, contract=contract)
        assert torch.equal(got, base * s), p
        lin = ops.linear(x * s, w, contract=contract)
        assert torch.equal(lin, ops.linear(x, w, contract=contract) * s), p
    # non-finite inputs stay visible: a NaN or Inf in a row reaches that row's outputs (and its neighbours')
    xb = x.clone()
    xb[5, 3] = float("nan")
    xb[9, 7] = float("inf")
    out = ops.layer_fused(g, xb, w, None, relu=False, contract=contract)
    assert not torch.isfinite(out[5]).any() and not torch.isfinite(out[9]).any()
    assert torch.isfinite(out).any()
