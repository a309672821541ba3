"""What a kernel of c2's size costs at all: a plain copy of [100 002, 64] fp32, an elementwise relu, rocBLAS x @ W^T, beside
this library's dense linear (K3 on K8's pipeline) and K2 / K4 -- back-to-back launches, event-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import ops
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
n = mesh.num_nodes
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), n)
x = torch.randn(n, 64, device=dev); w = torch.randn(64, 64, device=dev) / 8; b = torch.randn(64, device=dev)
y = torch.empty_like(x)
def t(name, fn, k=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:44s} {a.elapsed_time(e) / k * 1e3:6.1f} us")
t("empty launch (torch zero-size fill)", lambda: y[:1].zero_())
t("copy 25.6 MB -> 25.6 MB (y.copy_(x))", lambda: y.copy_(x))
t("relu out-of-place (torch.relu(x, out=y))", lambda: torch.relu(x, out=y) if False else torch.clamp_min(x, 0, out=y))
t("rocBLAS x @ W^T (torch.mm, fp32)", lambda: torch.mm(x, w.t(), out=y))
t("ops.linear 64 -> 64 (3xbf16)", lambda: ops.linear(x, w, b, contract="3xbf16"))
t("ops.linear 64 -> 64 (bf16x6)", lambda: ops.linear(x, w, b, contract="bf16x6"))
t("K2 propagate at 64", lambda: ops.propagate(g, x))
t("K4 layer 64 -> 64 (3xbf16)", lambda: ops.layer_fused(g, x, w, b, relu=True, contract="3xbf16"))
t("K4 layer 64 -> 64 (bf16x6)", lambda: ops.layer_fused(g, x, w, b, relu=True, contract="bf16x6"))
x16 = torch.randn(n, 16, device=dev); y16 = torch.empty_like(x16)
t("copy 6.4 MB -> 6.4 MB ([N, 16])", lambda: y16.copy_(x16))
t("K2 propagate at 16", lambda: ops.propagate(g, x16))
