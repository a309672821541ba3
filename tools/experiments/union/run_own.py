"""EXPERIMENT: own-rows-in-LDS gather vs K2 at 64 channels on the c2 mesh."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch, gwen_amd
from gwen_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libown.so"))
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="morton")
n = mesh.num_nodes
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), n)
gr, gc, gv = g.grouped()
assert gr is None
x = torch.randn(n, 64, device=dev)
out = torch.empty(n, 64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(mode):
    rc = lib.own_launch(mode, C.c_void_p(x.data_ptr()), C.c_void_p(gc.data_ptr()), C.c_void_p(gv.data_ptr()),
                        C.c_void_p(out.data_ptr()), n, st)
    assert rc == 0
ref = ops.propagate(g, x)
for mode in (0, 1):
    run(mode); torch.cuda.synchronize()
    print("mode", mode, "max abs diff vs K2:", float((out - ref).abs().max()))
for name, fn in (("own rows in LDS, conditional global loads", lambda: run(0)), ("same, unconditional global loads", lambda: run(1)),
                 ("K2", lambda: ops.propagate(g, x))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): fn()
    b_.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b_) / 100 * 1e3:.1f} us")
