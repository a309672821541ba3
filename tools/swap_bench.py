"""The one-line swap of INTEGRATION.md section 1 at the reference's own shape: GWEN's model composition
(six `GCNConv` calls + eager `torch.relu`, models_gnn.py:135-157,:189-212) built from gwen_amd.GCNConv, fed a
FRESH edge_index tensor per batch (as a NeighborLoader does) -- per-forward wall time, against the
whole-model path (gwen_amd.GNNModel, one C call).   python tools/swap_bench.py [members] [channels] [hidden]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125
c = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = "cuda:0"
torch.manual_seed(23)
G = gwen_amd.GCNConv
convs = [G(c, h), G(h, h // 2), G(h // 2, h // 4), G(h // 4, h // 2), G(h // 2, h), G(h, c)]
convs = [m.to(dev).eval() for m in convs]
ei_host = torch.from_numpy(gwen_amd.complete_graph(n))
x = torch.randn(n, c, device=dev)

def swapped(fresh):
    ei = ei_host.to(dev) if fresh else swapped.ei
    y = x
    for i, conv in enumerate(convs):
        y = conv(y, ei)
        if i < 5:
            y = torch.relu(y)
    return y
swapped.ei = ei_host.to(dev)

def timed(fn, k=30):
    with torch.no_grad():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6

model = gwen_amd.GNNModel(gwen_amd.GNNConfig(n, n, c, c, h)).to(dev).eval()
print(f"members={n} channels={c} hidden={h}")
print(f"  swapped GCNConv, fresh edge_index per forward : {timed(lambda: swapped(True)):8.1f} us")
print(f"  swapped GCNConv, one edge_index tensor        : {timed(lambda: swapped(False)):8.1f} us")
print(f"  gwen_amd.GNNModel (whole stack, one C call)   : {timed(lambda: model(x, swapped.ei)):8.1f} us")
print(f"  gwen_amd.GNNModel, fresh edge_index per forward: {timed(lambda: model(x, ei_host.to(dev))):8.1f} us")
