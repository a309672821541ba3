"""Training step of the c5-shaped InteractionNet forecaster (forward + MSE + backward + Adam) on the nu = 100 graphs:
ms per step with the product's backward (K6^T: atomic-free launches of libgwen_hip.so, gwen_amd/interaction.py)
and, for comparison, with round 2's backward -- the block restated in torch device ops (rocBLAS GEMMs,
index_select, index_add_ float atomics) and differentiated by autograd, kept HERE only as the "before" of that
comparison.      python tools/inet_train_bench.py [hidden] [blocks] [nu]   -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
from gwen_amd import interaction as I
from gwen_amd.forecaster import InteractionForecaster

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NU = int(sys.argv[3]) if len(sys.argv) > 3 else 100
C, dev = 8, "cuda:0"
mesh = gwen_amd.geodesic_mesh(NU, reorder="hilbert")
torch.manual_seed(23)
model = InteractionForecaster(C, H, S).to(dev).train()
graphs = model.prepare(mesh, dev)
x = torch.randn(mesh.faces.shape[0], C, device=dev)
y = torch.randn(mesh.faces.shape[0], C, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss = (model(x, graphs) - y).square().mean()
    loss.backward()
    opt.step()
    return loss


def timed(k=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k


def torch_backward(ctx, gx, ge):
    """Round 2's backward: recompute the block in torch device ops under autograd (library kernels, float atomics)."""
    x_src, x_dst, e = ctx.saved_tensors[:3]                  # ([3]: the forward's aggregate, [12:]: its node projections --
    params = ctx.saved_tensors[4:12]                         #  kept for the product's backward)
    net, graph, f = ctx.net, ctx.graph, ctx.net.channels
    act = {"none": lambda v: v, "relu": torch.relu, "silu": torch.nn.functional.silu}[net.activation]
    need = ctx.needs_input_grad[4:]
    with torch.enable_grad():
        xd = x_dst.detach().requires_grad_(need[1] or (ctx.same and need[0]))
        xs = xd if ctx.same else x_src.detach().requires_grad_(need[0])
        ee = e.detach().requires_grad_(need[2])
        w1, b1, w2, b2, w3, b3, w4, b4 = ps = [p.detach().requires_grad_(n) for p, n in zip(params, need[3:])]
        src, dst = graph.src.long(), graph.dst.long()
        pre = ee @ w1[:, :f].t() + (xs @ w1[:, f:2 * f].t()).index_select(0, src) + \
            (xd @ w1[:, 2 * f:].t() + b1).index_select(0, dst)
        m = act(pre) @ w2.t() + b2
        agg = torch.zeros_like(xd).index_add_(0, dst, m)
        if net.aggr == "mean":
            agg = agg / (graph.rowptr[1:] - graph.rowptr[:-1]).to(m.dtype).clamp(min=1).view(-1, 1)
        x_new = xd + act(xd @ w3[:, :f].t() + agg @ w3[:, f:].t() + b3) @ w4.t() + b4
        outs, gouts = [x_new], [gx]
        if ctx.update_edges and ge is not None and ge.numel():
            outs.append(ee + m); gouts.append(ge)
        wanted = [t for t in ([xd] if ctx.same else [xs, xd]) + [ee] + ps if t.requires_grad]
        grads = torch.autograd.grad(outs, wanted, gouts, allow_unused=True) if wanted else []
    it = iter(grads)
    take = lambda t: next(it) if t.requires_grad else None          # noqa: E731
    g_xs, g_xd = (None, take(xd)) if ctx.same else (take(xs), take(xd))
    g_e = take(ee)
    return (None, None, None, None, g_xs, g_xd, g_e, *[take(p) for p in ps])


t_new = timed()


def graphed_step_ms():
    """The whole step (forward, loss, backward, fused capturable Adam) captured once into a hipGraph and replayed: every
    launcher of the library is capturable, so what remains is the kernels without the ~350 launch gaps of a step."""
    opt_g = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up off the default stream, as capture requires
        for _ in range(3):
            opt_g.zero_grad(set_to_none=True)
            (model(x, graphs) - y).square().mean().backward()
            opt_g.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt_g.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss = (model(x, graphs) - y).square().mean()
        loss.backward()
        opt_g.step()
    g.replay()
    torch.cuda.synchronize()
    first = float(loss.detach())
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    assert float(loss.detach()) < first, (first, float(loss.detach()))       # the replayed step trains
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3


try:
    t_graph = float("nan") if os.environ.get("INET_SKIP_OLD") == "1" else round(graphed_step_ms(), 3)
except Exception as exc:                                # capture is best effort: say why it did not work
    t_graph = f"not captured: {type(exc).__name__}: {str(exc)[:200]}"
if os.environ.get("INET_SKIP_OLD") == "1":           # (profiling runs: only the product's kernels in the trace)
    t_old = float("nan")
else:
    keep = I._InteractionNetFunction.backward
    I._InteractionNetFunction.backward = staticmethod(torch_backward)
    t_old = timed()
    I._InteractionNetFunction.backward = keep
with torch.no_grad():
    model.eval()
    for _ in range(3):
        model(x, graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model(x, graphs)
    torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / 10
print(json.dumps({"workload": f"InteractionNet forecaster training step, nu={NU}, hidden {H}, {S} processor blocks, 1 member",
                  "train_step_ms_k6t_backward": round(t_new * 1e3, 3),
                  "train_step_ms_replayed_from_a_hipgraph": t_graph,
                  "train_step_ms_torch_recompute_backward_round2": round(t_old * 1e3, 3),
                  "inference_forward_ms": round(t_fwd * 1e3, 3)}))
