"""BASELINE config c3: nu = 100 mesh, F hidden channels (default 256), S chained F -> F GCN layers with ReLU
(default 4 "processor steps"), 1 member.  One JSON line: us per step sequence, edges/s per pass, roofline
of one pass with SURVEY 8(d)'s algorithmic bytes.   python tools/c3_bench.py [F] [S]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
n, e = mesh.num_nodes, mesh.num_edges
torch.manual_seed(23)
layers = []
for _ in range(S):
    conv = gwen_amd.GCNConv(F, F).to(dev)
    layers.append((conv.lin.weight.detach(), conv.bias.detach(), True, "auto"))
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), n)
plan = gwen_amd.StackForward(layers, g)
x = torch.randn(n, F, device=dev)
out = plan.run(x)
for _ in range(5):
    plan.run(x, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 50
for _ in range(K):
    plan.run(x, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
b_alg = 4 * F * (e + 2 * n) + 8 * e + 8 * n
per_pass = dt / S
print(json.dumps({"workload": f"c3: nu=100 N={n} E={e}, {S} chained GCN layers {F}->{F} + ReLU, 1 member",
                  "us_per_sequence": round(dt * 1e6, 1), "us_per_pass": round(per_pass * 1e6, 1),
                  "edges_per_s_per_pass": round(e / per_pass),
                  "roofline": {"bound": "hbm", "algorithmic_bytes_per_pass": b_alg,
                               "achieved_GBs": round(b_alg / per_pass / 1e9, 1), "peak_GBs": 8000.0,
                               "frac": round(b_alg / per_pass / 8e12, 3)}}))
