"""K9 timing at nu=100: the chain (32,16,32 pre-projected) against the kernels it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import ops
dev = "cuda:0"
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 100
members = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = gwen_amd.geodesic_mesh(nu, reorder="hilbert")
n = m.num_nodes
g = gwen_amd.prepare_graph(torch.from_numpy(m.edge_index).to(dev), n)
torch.manual_seed(3)
x = torch.randn(members, n, 32, device=dev) if members > 1 else torch.randn(n, 32, device=dev)
st = [(None, torch.randn(32, device=dev) * .1, True), (torch.randn(16, 32, device=dev) * .2, torch.randn(16, device=dev) * .1, True),
      (torch.randn(32, 16, device=dev) * .2, torch.randn(32, device=dev) * .1, True)]
print("hops", None if g.hops(3) is None else "ok")
def timed(fn, k=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
print("K9 chain us", round(timed(lambda: ops.narrow_chain(g, x, st)), 2))
print("K9 2 stages us", round(timed(lambda: ops.narrow_chain(g, x, st[:2])), 2))
print("K9 1 stage us", round(timed(lambda: ops.narrow_chain(g, x, st[:1])), 2))
