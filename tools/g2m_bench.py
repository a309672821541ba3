"""Timing of the build-defined grid->mesh->grid forecaster (BASELINE config c5 shape): geodesic mesh
nu = 100 (100 002 vertices, 200 000 grid cells), 4 processor steps.  python tools/g2m_bench.py [C] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
from gwen_amd import g2m
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
torch.manual_seed(23)
model = g2m.GridMeshGridModel(C, H, 4).to(dev).eval()
graphs = model.prepare(mesh, dev)
x = torch.randn(mesh.faces.shape[0], C, device=dev)
with torch.no_grad():
    for _ in range(3):
        model(x, graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        y = model(x, graphs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    states = model.rollout(x, graphs, 4)
    torch.cuda.synchronize()
    dr = time.perf_counter() - t0
edges = 2 * 600000 + 4 * 600000
print(f"C={C} H={H}: forward {dt*1e6:.0f} us ({edges/dt/1e9:.2f} G edge-passes/s), 4-step rollout {dr*1e3:.2f} ms")
