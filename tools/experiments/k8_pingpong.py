"""K8 256->256 x M members: one layer repeated on the same (x, out) against ping-pong between two buffers (what a
stack does): is the in-stack slowdown a buffer-alternation effect?  python tools/experiments/k8_pingpong.py [F] [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).cuda(), mesh.num_nodes)
n = mesh.num_nodes
w = torch.randn(F, F, device="cuda") / F ** .5; b = torch.randn(F, device="cuda") * .1
bufs = [torch.randn(M, n, F, device="cuda") for _ in range(5)]
def timed(fn, k=24):
    for _ in range(4): fn(0)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(k): fn(i)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / k * 1e3
from gwen_amd import _lib
from gwen_amd.graph import _ptr, _stream
tr, tl, tv, umax = g.tiles()
L = _lib.lib()
def layer(src, dst):
    rc = L.gwen_gcn_wide_layer_f32(_ptr(tr), _ptr(tl), _ptr(tv), _ptr(src), _ptr(w), _ptr(b), _ptr(dst), n, n, F, F, F, M,
                                   n * F, n * F, 1, umax, 0, _stream(torch.device("cuda:0")))
    assert rc == 0
print("same buffers      ", round(timed(lambda i: layer(bufs[0], bufs[1])), 1), "us")
print("ping-pong 2 bufs  ", round(timed(lambda i: layer(bufs[i & 1], bufs[1 - (i & 1)])), 1), "us")
print("rotate 3 bufs     ", round(timed(lambda i: layer(bufs[i % 3], bufs[(i + 1) % 3])), 1), "us")
print("rotate 4 bufs     ", round(timed(lambda i: layer(bufs[i % 4], bufs[(i + 1) % 4])), 1), "us")
print("rotate 5 bufs     ", round(timed(lambda i: layer(bufs[i % 5], bufs[(i + 1) % 5])), 1), "us")
print("rotate 3 bufs     ", round(timed(lambda i: layer(bufs[i % 3], bufs[(i + 1) % 3])), 1), "us")
print("ping-pong 2 bufs  ", round(timed(lambda i: layer(bufs[i & 1], bufs[1 - (i & 1)])), 1), "us")
