"""Timing of the InteractionNet encode-process-decode forecaster (BASELINE config c5 shape): geodesic mesh
nu = 100 (100 002 vertices, 200 000 grid cells), 4 processor blocks, 4-step autoregressive rollout of M local
members (c5: 32 members over 8 GPUs = 4 per GPU) -- all members through ONE launch set per step (block-diagonal
graph) against one member after the other.
python tools/forecaster_bench.py [grid_channels] [hidden] [steps] [members]   -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
from gwen_amd.forecaster import GraphedStep, InteractionForecaster, ensemble_forecast
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
M = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
torch.manual_seed(23)
model = InteractionForecaster(C, H, S).to(dev).eval()
graphs = model.prepare(mesh, dev)
x = torch.randn(mesh.faces.shape[0], C, device=dev)
xm = torch.randn(M, mesh.faces.shape[0], C, device=dev)


def timed(fn, k=10, warm=2):
    t0 = time.perf_counter()             # >= 0.1 s of the same work first: the clocks ramp for tens of ms after idling
    while time.perf_counter() - t0 < 0.1:
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k


with torch.no_grad():
    dt = timed(lambda: model(x, graphs))
    gs = GraphedStep(model, graphs, x)
    dg = timed(lambda: gs(x))
    dr = timed(lambda: model.rollout(x, graphs, 4), k=3, warm=1)
    cache = {}          # the captured steps live here: the timed calls replay only
    d_loop = timed(lambda: ensemble_forecast(model, graphs, xm, 4, M, graphed=True, batched=False, step_cache=cache), k=3, warm=1)
    d_batch = timed(lambda: ensemble_forecast(model, graphs, xm, 4, M, graphed=True, batched=True, step_cache=cache), k=3, warm=1)
    same = torch.equal(ensemble_forecast(model, graphs, xm, 4, M, batched=False), ensemble_forecast(model, graphs, xm, 4, M))
edges = graphs.g2m.num_edges + S * graphs.mesh.num_edges + graphs.m2g.num_edges
print(json.dumps({"workload": f"InteractionNet forecaster nu=100 grid={mesh.faces.shape[0]} mesh={mesh.num_nodes} "
                              f"C={C} H={H} processor_blocks={S}", "forward_us": round(dt * 1e6, 1),
                  "graphed_forward_us": round(dg * 1e6, 1), "edge_updates_per_s": round(edges / dt),
                  "rollout4_ms_one_member": round(dr * 1e3, 3),
                  "members": M, "rollout4_ms_members_one_by_one": round(d_loop * 1e3, 3),
                  "rollout4_ms_members_batched": round(d_batch * 1e3, 3),
                  "members_per_s_4step_batched": round(M / d_batch, 2),
                  "edge_updates_per_s_batched": round(4 * M * edges / d_batch), "batched_equals_loop_bitwise": bool(same)}))
