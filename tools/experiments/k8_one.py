"""One K8 layer, timed:  python tools/experiments/k8_one.py FIN FOUT MEMBERS [3xbf16|bf16x6|f16x3]  (nu = 100 Hilbert mesh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import ops
fin, fout, M = (int(v) for v in sys.argv[1:4])
contract = sys.argv[4] if len(sys.argv) > 4 else "3xbf16"
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), mesh.num_nodes)
torch.manual_seed(3)
x = torch.randn(M, mesh.num_nodes, fin, device=dev)
w = torch.randn(fout, fin, device=dev) / fin ** 0.5
b = torch.randn(fout, device=dev)
out = ops.wide_layer(g, x, w, b, relu=True, contract=contract)
for _ in range(30):
    ops.wide_layer(g, x, w, b, relu=True, contract=contract)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K = 50
for _ in range(K):
    ops.wide_layer(g, x, w, b, relu=True, contract=contract)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / K * 1e3
comp = 4 * M * mesh.num_nodes * (fin + fout)
print(f"K8 {fin}->{fout} x {M} members {contract}: {us:.1f} us  {comp / us / 1e6:.2f} TB/s compulsory  checksum {out.double().sum().item():.10e}")
