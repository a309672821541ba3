"""BASELINE config c3: nu = 100 mesh, F hidden channels (default 256), S chained F -> F GCN layers with ReLU
(default 4 "processor steps"), M members (default 1; 4 = config c5's per-GPU load).  One JSON line: us per
sequence and per pass, edges/s per pass, the roofline of one pass (compulsory bytes / time / 8 TB/s, and
SURVEY 8(d)'s L2-path bytes beside it), and the relative error of one member against the plain-C oracle
(oracle/gcn_ref.c, fp64) chained on the host.     python tools/c3_bench.py [F] [S] [M] [--precision f16x3|bf16x6|3xbf16]
(default: the library default f16x3 -- K8 on the scaled fp16 split, one launch per layer; bf16x6 -- K8 as two
256 -> 128 launches at 256 channels; 3xbf16 -- K8 on the 17-bit tier)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, gwen_amd
PREC = "f16x3"
if "--precision" in sys.argv:
    i = sys.argv.index("--precision")
    PREC = sys.argv[i + 1]
    del sys.argv[i:i + 2]
ORDER = {"f16x3": "auto", "bf16x6": "auto_x6", "3xbf16": "auto_x3"}[PREC]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
n, e = mesh.num_nodes, mesh.num_edges
torch.manual_seed(23)
layers = []
for _ in range(S):
    conv = gwen_amd.GCNConv(F, F).to(dev)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    layers.append((conv.lin.weight.detach(), conv.bias.detach(), True, ORDER))
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), n)
plan = gwen_amd.StackForward(layers, g)
x = torch.randn(M, n, F, device=dev) if M > 1 else torch.randn(n, F, device=dev)
ev = gwen_amd.KernelEvents(2 * S)
out = plan.run(x, events=ev)
kinds = sorted({k for k, *_ in ev.durations()})
t0 = time.perf_counter()                 # 0.1 s of the same work first: the clocks ramp for tens of ms after idling
while time.perf_counter() - t0 < 0.1:
    for _ in range(5):
        plan.run(x, out=out)
    torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K):
    plan.run(x, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
per_pass = dt / S
comp = 8 * M * n * F + 8 * (e + n) + 4 * n + 4 * F * F
b_l2 = M * (4 * F * (e + 2 * n) + 8 * e + 8 * n)
rec = {"workload": f"c3: nu=100 N={n} E={e}, {S} chained GCN layers {F}->{F} + ReLU, {M} member(s)", "precision": PREC,
       "kernels": kinds, "us_per_sequence": round(dt * 1e6, 1), "us_per_pass": round(per_pass * 1e6, 1),
       "edges_per_s_per_pass": round(M * e / per_pass),
       "roofline": {"bound": "hbm" if 8 * M * n * F > 256 * 2 ** 20 else "l2/infinity-cache (in+out fit 256 MiB)",
                    "compulsory_bytes_per_pass": comp, "achieved_GBs": round(comp / per_pass / 1e9, 1),
                    "peak_GBs": 8000.0, "frac": round(comp / per_pass / 8e12, 3),
                    "l2_path_bytes_per_pass_survey_8d": b_l2}}
if "--no-parity" not in sys.argv:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import gcn_ref
    gcn_ref.build()
    x0 = (x[0] if M > 1 else x).cpu().numpy()
    want = x0.astype(np.float64)
    for w, b, _, _ in layers:
        want = gcn_ref.conv(want.astype(np.float32), mesh.edge_index, w.cpu().numpy(), b.cpu().numpy(), relu=True, f64=True)
    got = (out[0] if M > 1 else out).cpu().numpy()
    rec["rel_err_vs_c_oracle_fp64"] = float(np.abs(got - want).max() / np.abs(want).max())
print(json.dumps(rec))
