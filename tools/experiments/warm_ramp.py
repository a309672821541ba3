"""Does the c2 step time depend on how long the GPU has been busy?  Times consecutive groups of 50 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
n = mesh.num_nodes
torch.manual_seed(23)
model = gwen_amd.GNNModel(gwen_amd.GNNConfig(n, n, 64, 64, 64)).to(dev).eval()
x = torch.randn(n, 64, device=dev)
ei = torch.from_numpy(mesh.edge_index).to(dev)
with torch.no_grad():
    model(x, ei); torch.cuda.synchronize()
    time.sleep(2.0)                       # idle: clocks fall back
    out = []
    t_start = time.perf_counter()
    for grp in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): model(x, ei)
        b.record(); torch.cuda.synchronize()
        out.append((round((time.perf_counter() - t_start) * 1e3, 1), round(a.elapsed_time(b) / 50 * 1e3, 1)))
print("elapsed ms, us/step:", out)
