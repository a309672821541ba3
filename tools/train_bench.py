"""Training step (forward + L1 loss on masked rows + backward + Adam) of GNNModel on the c2 mesh and on the
reference's own shape: ms per step.   python tools/train_bench.py [mesh|mesh256|ref] [fused] [graph]
fused: torch.optim.Adam(fused=True);  graph: the WHOLE step (forward, loss, backward, Adam) captured once into a
hipGraph (torch.cuda.CUDAGraph) and replayed -- every launcher of libgwen_hip.so is capturable (no allocation, no
synchronisation inside), so the step leaves the host's per-launch cost behind."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
which = sys.argv[1] if len(sys.argv) > 1 else "mesh"
dev = "cuda:0"
torch.manual_seed(23)
if which in ("mesh", "mesh256"):                   # c2's model / the same stack at c3's 256 channels
    mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
    n, c, h = mesh.num_nodes, (64 if which == "mesh" else 256), (64 if which == "mesh" else 256)
    ei = torch.from_numpy(mesh.edge_index).to(dev)
else:
    n, c, h = 125, 16384, 1024
    ei = torch.from_numpy(gwen_amd.complete_graph(n)).to(dev)
model = gwen_amd.GNNModel(gwen_amd.GNNConfig(n, n, c, c, h)).to(dev).train()
GRAPH = "graph" in sys.argv
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=("fused" in sys.argv) or GRAPH,
                       capturable=GRAPH)   # the reference: the default (foreach)
x = torch.randn(n, c, device=dev)
mask = torch.rand(n, device=dev) < 0.5
def step():
    opt.zero_grad(set_to_none=True)
    out = model(x, ei)
    loss = gwen_amd.loss_func(out, x, mask)
    loss.backward()
    opt.step()
    return loss
K = 30
if GRAPH:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up off the default stream, as capture requires
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        out = model(x, ei)
        loss = gwen_amd.loss_func(out, x, mask)
        loss.backward()
        opt.step()
    g.replay()
    torch.cuda.synchronize()
    first = float(loss.detach())
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    assert float(loss.detach()) < first, (first, float(loss.detach()))   # the replayed step trains
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
else:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
with torch.no_grad():
    model.eval()
    for _ in range(3):
        model(x, ei)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        model(x, ei)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / K
print(("whole step replayed from a hipGraph, " if GRAPH else "") + ("fused Adam, " if "fused" in sys.argv else "") + f"{which}: N={n} C={c} H={h}: training step {dt*1e3:.3f} ms, inference forward {df*1e3:.3f} ms")
