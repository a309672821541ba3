"""EXPERIMENT: the union-staged layer (layer_u.hip) against K4, bitwise, and timed (c2 mesh, 64 -> 64)."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch, gwen_amd
from gwen_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, sys.argv[1] if len(sys.argv) > 1 else "liblayer_u.so"))
RB, HC, NS = lib.union2_rb(), lib.union2_hc(), lib.union2_ns()
contract = {2: "3xbf16", 3: "bf16x6"}[NS]
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
n = mesh.num_nodes
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), n)
gr, gc, gv = g.grouped()
assert gr is None
col = gc.cpu().numpy()[:8 * n].reshape(n, 8).astype(np.int64)
assert col.max() < n
nb = (n + RB - 1) // RB
lid = np.zeros((n, 8), dtype=np.uint16)
halo = np.zeros((nb, HC), dtype=np.int32)
mx = 0
for b in range(nb):
    r0, r1 = RB * b, min(RB * b + RB, n)
    c = col[r0:r1]
    inside = (c >= r0) & (c < r1)
    ext = np.unique(c[~inside])
    mx = max(mx, len(ext))
    assert len(ext) <= HC, (len(ext), HC)
    halo[b, :len(ext)] = ext
    halo[b, len(ext):] = r0
    lid[r0:r1] = np.where(inside, c - r0, RB + np.searchsorted(ext, c)).astype(np.uint16)
print("RB", RB, "blocks", nb, "max halo", mx, contract)
torch.manual_seed(5)
x = torch.randn(n, 64, device=dev)
w = torch.randn(64, 64, device=dev) / 8
bias = torch.randn(64, device=dev)
lid_t = torch.from_numpy(lid).to(dev); halo_t = torch.from_numpy(halo).to(dev)
out = torch.empty(n, 64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
def run(relu=1):
    rc = lib.layer_u_launch(P(x), P(gv), P(lid_t), P(halo_t), P(w), P(bias), P(out), n, relu, st)
    assert rc == 0
for relu in (1, 0):
    run(relu); torch.cuda.synchronize()
    ref = ops.layer_fused(g, x, w, bias, relu=bool(relu), contract=contract)
    print("relu", relu, "bitwise K4:", bool(torch.equal(out, ref)), "max abs diff", float((out - ref).abs().max()))
for name, fn in (("layer_u", run), ("K4", lambda: ops.layer_fused(g, x, w, bias, relu=True, contract=contract))):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): fn()
    b_.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b_) / 200 * 1e3:.1f} us")
