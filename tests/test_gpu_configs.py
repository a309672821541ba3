"""Parity at BASELINE.json's config sizes, through the same entry points bench.py uses.

c2: the whole GNNModel(64, 64) forward (stack launcher: K5 chains + K4 layers) on the nu = 100 mesh against
    OracleGNNModel on the CPU -- the kernels of the headline bench at the headline size.
c3: 4 chained 256 -> 256 GCN layers + ReLU on the same mesh (K8 through the stack launcher) against the plain-C
    oracle (oracle/gcn_ref.c, fp64) chained on the host.
c5 shape: one member through one step of the InteractionNet forecaster (grid -> mesh -> grid, 4 processor
    steps) on the full mesh: 64 channels against the fp64 torch oracle, 256 channels through the closed
    forms (zero read-out / zero second layers leave the state untouched) and determinism.
The reference functions on this path: GNNModel.forward /root/reference/src/gwen/models_gnn.py:292-303
(:135-157, :189-212); c3 / c5 are BASELINE-defined workloads."""
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


def row_rel_err(got, want, floor=1e-3):
    """Largest per-row error relative to that row's own largest magnitude (rows of small magnitude cannot hide
    behind the tensor's maximum; `floor` keeps all-zero ReLU rows finite)."""
    g, w = torch.as_tensor(np.asarray(got.cpu())).double(), torch.as_tensor(np.asarray(want)).double()
    return float(((g - w).abs().amax(-1) / w.abs().amax(-1).clamp_min(floor)).max())


@pytest.mark.parametrize("reorder", ["hilbert", "morton"])
def test_c2_whole_model_at_config_size(ga, reorder):
    from oracle import gcn_oracle as O
    m = ga.geodesic_mesh(100, reorder=reorder)
    n = m.num_nodes
    assert (n, m.num_edges) == (100002, 600000)
    ei = torch.from_numpy(m.edge_index)
    torch.manual_seed(SEED)
    ref = O.OracleGNNModel(O.OracleGNNConfig(n, n, 64, 64, 64))
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.1)
    model = ga.GNNModel(ga.GNNConfig(n, n, 64, 64, 64))
    model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(DEV).eval()
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(SEED))
    g = model.prepare(ei.to(DEV), n)
    plan = ga.StackForward(model.stack(), g)
    ev = ga.KernelEvents(12)
    got = plan.run(x.to(DEV), events=ev)
    kinds = [k for k, *_ in ev.durations()]
    assert kinds.count("chain") == 3 and kinds.count("layer") == 3          # the bench's kernels
    with torch.no_grad():
        want = ref(x, ei)
        assert torch.equal(model(x.to(DEV), g), got)
    assert rel_err(got, want) <= REL_TOL
    assert row_rel_err(got, want) <= 10 * REL_TOL
    # the exact-fp32 orders of the same model
    for mod in model.modules():
        if isinstance(mod, ga.GCNConv):
            mod.order = "fused_exact"
    with torch.no_grad():
        exact = model(x.to(DEV), g)
    assert rel_err(exact, want) <= 2e-6


def test_c3_processor_stack_at_config_size(ga, cref):
    m = ga.geodesic_mesh(100, reorder="hilbert")
    n = m.num_nodes
    ei = torch.from_numpy(m.edge_index)
    F, steps = 256, 4
    params = [make_params(F, F, seed=SEED + k) for k in range(steps)]
    x = torch.randn(n, F, generator=torch.Generator().manual_seed(SEED))
    want = x.numpy().astype(np.float64)
    for w, b in params:            # fp64 accumulation of fp32 inputs, layer by layer, then back to fp32 as the device does
        want = cref.conv(want.astype(np.float32), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
    g = ga.prepare_graph(ei.to(DEV), n)
    # precision "3xbf16": K8, the HBM-leg kernel of bench.py (three images of W exceed its registers at 256 channels)
    plan = ga.StackForward([(w.to(DEV), b.to(DEV), True, "auto_x3") for w, b in params], g)
    ev = ga.KernelEvents(2 * steps)
    got = plan.run(x.to(DEV), events=ev)
    assert [k for k, *_ in ev.durations()] == ["wide"] * steps
    assert rel_err(got, want) <= REL_TOL
    assert row_rel_err(got, want) <= 10 * REL_TOL
    assert torch.equal(got, plan.run(x.to(DEV)))
    # the default precision (fp32-class; K8 on the scaled fp16 split, ONE launch per layer) and precision "bf16x6" (K8
    # too: two 256 -> 128 launches per layer -- three bf16 images of W for all 256 columns exceed a wave's registers):
    # an order of magnitude closer to fp64
    for order in ("auto", "auto_x6"):
        plan6 = ga.StackForward([(w.to(DEV), b.to(DEV), True, order) for w, b in params], g)
        got6 = plan6.run(x.to(DEV), events=ev)
        assert [k for k, *_ in ev.durations()] == ["wide"] * steps
        assert rel_err(got6, want) <= 2e-6 and rel_err(got6, want) < rel_err(got, want), order
        assert row_rel_err(got6, want) <= 2e-5, order
        assert torch.equal(got6, plan6.run(x.to(DEV)))


def _forecaster_inputs(ga, nu, C, H, steps):
    from gwen_amd.forecaster import InteractionForecaster
    m = ga.geodesic_mesh(nu, reorder="hilbert")
    torch.manual_seed(SEED)
    model = InteractionForecaster(C, H, steps)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    x0 = torch.randn(m.faces.shape[0], C, generator=torch.Generator().manual_seed(SEED))
    return m, model, x0


def test_c5_forecaster_step_64ch_at_config_size_vs_fp64_oracle(ga):
    from gwen_amd import g2m
    from gwen_amd.forecaster import InteractionForecaster, edge_features
    from oracle import interaction_oracle as IO
    C, H, steps = 8, 64, 4
    m, model, x0 = _forecaster_inputs(ga, 100, C, H, steps)
    assert m.faces.shape[0] == 200000
    a, b = g2m.grid_mesh_edges(m)
    cell = m.pos[m.faces].mean(axis=1)
    cell /= np.linalg.norm(cell, axis=1, keepdims=True)
    f = [torch.from_numpy(v).double() for v in (edge_features(cell, m.pos, a), edge_features(m.pos, m.pos, m.edge_index),
                                                 edge_features(m.pos, cell, b))]
    sd = {k: v.double() for k, v in model.state_dict().items()}
    want = IO.forecaster_step(sd, x0.double(), torch.from_numpy(m.pos.astype(np.float32)).double(), torch.from_numpy(a),
                              torch.from_numpy(m.edge_index), torch.from_numpy(b), *f, steps)
    graphs = InteractionForecaster.prepare(m, DEV)
    model = model.to(DEV)
    with torch.no_grad():
        got = model(x0.to(DEV), graphs)
        assert torch.equal(got, model(x0.to(DEV), graphs))
    assert rel_err(got, want) <= REL_TOL


def test_c5_forecaster_step_256ch_at_config_size_closed_forms(ga):
    from gwen_amd.forecaster import InteractionForecaster
    C, H, steps = 8, 256, 4
    m, model, x0 = _forecaster_inputs(ga, 100, C, H, steps)
    graphs = InteractionForecaster.prepare(m, DEV)
    model = model.to(DEV)
    xd = x0.to(DEV)
    with torch.no_grad():
        y = model(xd, graphs)
        assert torch.isfinite(y).all() and torch.equal(y, model(xd, graphs))
        # zero read-out: the step is the identity on the grid state, whatever the 6 blocks did
        w_keep, b_keep = model.readout.weight.clone(), model.readout.bias.clone()
        model.readout.weight.zero_(); model.readout.bias.zero_()
        assert torch.equal(model(xd, graphs), xd)
        model.readout.weight.copy_(w_keep); model.readout.bias.copy_(b_keep)
        # zero second layers in every block: node and edge states pass through unchanged, so the output is
        # grid_x + readout(grid_embed(grid_x)) -- two dense projections, checked in fp64 on the host
        for blk in [model.encoder, *model.processor, model.decoder]:
            for mlp in (blk.edge_mlp, blk.node_mlp):
                mlp[2].weight.zero_(); mlp[2].bias.zero_()
        got = model(xd, graphs)
    with torch.no_grad():
        vg = x0.double() @ model.grid_embed.weight.double().cpu().t() + model.grid_embed.bias.double().cpu()
        want = x0.double() + vg @ w_keep.double().cpu().t() + b_keep.double().cpu()
    assert rel_err(got, want) <= REL_TOL


def test_c5_rollout_four_members_four_steps(ga):
    """BASELINE configs[4] per GPU: 4 members x 4 autoregressive steps through ``ensemble_forecast`` (all members in
    one launch set per step, the step captured once and replayed) on the nu = 100 graphs.  64 channels: one member's
    whole trajectory against the fp64 oracle stepped four times; 256 channels (the config's width; an fp64 oracle
    rollout there is many minutes of CPU): bitwise determinism, batched == member by member, and the closed form of
    a zero read-out (every step is the identity on the grid state)."""
    from gwen_amd import g2m
    from gwen_amd.forecaster import InteractionForecaster, edge_features, ensemble_forecast
    from oracle import interaction_oracle as IO
    members, n_steps, blocks = 4, 4, 4
    # ---- 64 channels vs the oracle -------------------------------------------------------------------------------
    C, H = 8, 64
    m, model, _ = _forecaster_inputs(ga, 100, C, H, blocks)
    n_grid = m.faces.shape[0]
    xs = torch.stack([torch.randn(n_grid, C, generator=torch.Generator().manual_seed(SEED + k)) * 0.5
                      for k in range(members)])
    a, b = g2m.grid_mesh_edges(m)
    cell = m.pos[m.faces].mean(axis=1)
    cell /= np.linalg.norm(cell, axis=1, keepdims=True)
    f = [torch.from_numpy(v).double() for v in (edge_features(cell, m.pos, a), edge_features(m.pos, m.pos, m.edge_index),
                                                 edge_features(m.pos, cell, b))]
    sd = {k: v.double() for k, v in model.state_dict().items()}
    pos64 = torch.from_numpy(m.pos.astype(np.float32)).double()
    pick = 2
    want = xs[pick].double()
    for _ in range(n_steps):
        want = IO.forecaster_step(sd, want, pos64, torch.from_numpy(a), torch.from_numpy(m.edge_index),
                                  torch.from_numpy(b), *f, blocks)
    graphs = InteractionForecaster.prepare(m, DEV)
    model = model.to(DEV).eval()
    cache = {}
    got = ensemble_forecast(model, graphs, xs.to(DEV), n_steps, members, graphed=True, batched=True, step_cache=cache)
    assert got.shape == (members, n_grid, C) and torch.isfinite(got).all()
    assert rel_err(got[pick], want) <= REL_TOL
    assert torch.equal(got, ensemble_forecast(model, graphs, xs.to(DEV), n_steps, members, graphed=True, batched=True,
                                              step_cache=cache))
    del graphs, model, cache
    torch.cuda.empty_cache()
    # ---- 256 channels: the config's width -----------------------------------------------------------------------------
    H = 256
    m, model, _ = _forecaster_inputs(ga, 100, C, H, blocks)
    graphs = InteractionForecaster.prepare(m, DEV)
    model = model.to(DEV).eval()
    xd = xs.to(DEV)
    cache = {}
    got = ensemble_forecast(model, graphs, xd, n_steps, members, graphed=True, batched=True, step_cache=cache)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ensemble_forecast(model, graphs, xd, n_steps, members, graphed=True, batched=True,
                                              step_cache=cache))                        # replayed: same bits
    one_by_one = ensemble_forecast(model, graphs, xd, n_steps, members, graphed=False, batched=False)
    assert torch.equal(got, one_by_one)                                                 # block-diagonal batch == loop
    with torch.no_grad():
        model.readout.weight.zero_(); model.readout.bias.zero_()
    assert torch.equal(ensemble_forecast(model, graphs, xd, n_steps, members, graphed=False, batched=True), xd)
