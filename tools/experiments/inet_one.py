"""One K6 variant in isolation, for rocprofv3 --pmc runs: python inet_one.py <F> <mode> (plain|gather|edge)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd.interaction import interaction_graph, mlp2
F, mode = int(sys.argv[1]), sys.argv[2]
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100)
g = interaction_graph(torch.from_numpy(mesh.edge_index).to(dev), mesh.num_nodes, mesh.num_nodes)
torch.manual_seed(0)
e = torch.randn(g.num_edges, F, device=dev)
w1, w2 = torch.randn(F, F, device=dev) / F ** 0.5, torch.randn(F, F, device=dev) / F ** 0.5
ps, pd = torch.randn(mesh.num_nodes, F, device=dev), torch.randn(mesh.num_nodes, F, device=dev)
b = torch.randn(F, device=dev)
for _ in range(5):
    if mode == "plain":
        mlp2(e, w1, w2, b, res=e)
    elif mode == "gather":
        mlp2(e, w1, w2, b, g1=ps, idx1=g.src, g2=pd, idx2=g.dst, res=e)
    else:
        mlp2(e, w1, w2, b, g1=ps, idx1=g.src, g2=pd, idx2=g.dst, res=e, graph=g)
torch.cuda.synchronize()
