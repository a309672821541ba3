import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
from gwen_amd import ops
mesh = gwen_amd.geodesic_mesh(100, reorder="morton")
N = mesh.num_nodes
ei = torch.from_numpy(mesh.edge_index).cuda()
g = gwen_amd.prepare_graph(ei, N)
x32 = torch.randn(N, 32, device="cuda"); w = torch.randn(16, 32, device="cuda") * 0.2
b32 = torch.randn(32, device="cuda"); b16 = torch.randn(16, device="cuda")
h16 = torch.randn(N, 16, device="cuda")
mode = sys.argv[1]
for _ in range(200):
    if mode == "alone":
        ops.propagate(g, h16, b16, True)
    elif mode == "after_chain":
        h = ops.chain(g, x32, w, None, b32, True, True)
        ops.propagate(g, h, b16, True)
    elif mode == "after_k4":
        h = ops.layer_fused(g, x32, w, b16, True)
        ops.propagate(g, h, b16, True)
    elif mode == "zeros":
        ops.propagate(g, torch.zeros_like(h16), b16, True)
    elif mode == "small":
        ops.propagate(g, h16 * 1e-3, b16 * 0, True)
    elif mode == "denorm":
        ops.propagate(g, h16 * 1e-40, b16 * 0, True)
    elif mode == "after_copy":
        h = h16.clone()
        ops.propagate(g, h, b16, True)
torch.cuda.synchronize()
