"""Timing of the InteractionNet encode-process-decode forecaster (BASELINE config c5 shape, one member):
geodesic mesh nu = 100 (100 002 vertices, 200 000 grid cells), 4 processor blocks, 4-step rollout.
python tools/forecaster_bench.py [grid_channels] [hidden] [steps]   -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
from gwen_amd.forecaster import InteractionForecaster
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="morton")
torch.manual_seed(23)
model = InteractionForecaster(C, H, S).to(dev).eval()
graphs = model.prepare(mesh, dev)
x = torch.randn(mesh.faces.shape[0], C, device=dev)
with torch.no_grad():
    for _ in range(3):
        model(x, graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model(x, graphs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    model.rollout(x, graphs, 4)
    torch.cuda.synchronize()
    dr = time.perf_counter() - t0
    from gwen_amd.forecaster import GraphedStep
    gs = GraphedStep(model, graphs, x)
    for _ in range(3):
        gs(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gs(x)
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / 10
edges = graphs.g2m.num_edges + S * graphs.mesh.num_edges + graphs.m2g.num_edges
print(json.dumps({"workload": f"InteractionNet forecaster nu=100 grid={mesh.faces.shape[0]} mesh={mesh.num_nodes} "
                              f"C={C} H={H} processor_blocks={S}", "forward_us": round(dt * 1e6, 1), "graphed_forward_us": round(dg * 1e6, 1),
                  "edge_updates_per_s": round(edges / dt), "rollout4_ms": round(dr * 1e3, 3),
                  "members_per_s_4step": round(1.0 / dr, 2)}))
