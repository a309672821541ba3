"""EXPERIMENT: LDS-staged union gather vs K2 at 64 channels on the c2 mesh."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch, gwen_amd
from gwen_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libunion.so"))
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="morton")
n = mesh.num_nodes
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), n)
gr, gc, gv = g.grouped()
assert gr is None
col = gc.cpu().numpy()[:8 * n].reshape(n, 8).astype(np.int64)
HC = 104
nb = (n + 63) // 64
lid = np.zeros((n, 8), dtype=np.uint8)
halo = np.zeros((nb, HC), dtype=np.int32)
mx = 0
for b in range(nb):
    r0, r1 = 64 * b, min(64 * b + 64, n)
    c = col[r0:r1]
    inside = (c >= r0) & (c < r0 + 64)
    ext = np.unique(c[~inside])
    mx = max(mx, len(ext))
    assert len(ext) <= HC
    halo[b, :len(ext)] = ext
    halo[b, len(ext):] = r0
    l = np.where(inside, c - r0, 64 + np.searchsorted(ext, c))
    lid[r0:r1] = l.astype(np.uint8)
print("max halo", mx)
x = torch.randn(n, 64, device=dev)
lid_t = torch.from_numpy(lid).to(dev); halo_t = torch.from_numpy(halo).to(dev)
out = torch.empty(n, 64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    rc = lib.union_launch(C.c_void_p(x.data_ptr()), C.c_void_p(gv.data_ptr()), C.c_void_p(lid_t.data_ptr()),
                          C.c_void_p(halo_t.data_ptr()), C.c_void_p(out.data_ptr()), n, st)
    assert rc == 0
run(); torch.cuda.synchronize()
ref = ops.propagate(g, x)
print("max abs diff vs K2:", float((out - ref).abs().max()))
for name, fn in (("union", run), ("K2", lambda: ops.propagate(g, x))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): fn()
    b_.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b_) / 100 * 1e3:.1f} us")
