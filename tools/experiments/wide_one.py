"""One K8 shape a few times (for rocprofv3 passes): python tools/experiments/wide_one.py [nu F M reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import ops
nu, F, M, reps = (int(v) for v in (sys.argv[1:5] + ["100", "256", "4", "6"][len(sys.argv) - 1:]))
mesh = gwen_amd.geodesic_mesh(nu, reorder=os.environ.get("KB_REORDER", "hilbert"))
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).cuda(), mesh.num_nodes)
x = torch.randn(M, mesh.num_nodes, F, device="cuda")
w = torch.randn(F, F, device="cuda") / F ** 0.5
b = torch.randn(F, device="cuda")
for _ in range(reps):
    out = ops.wide_layer(g, x, w, b, relu=True)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
