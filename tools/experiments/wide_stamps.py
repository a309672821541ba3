"""In-kernel s_memtime stamps of K8's pipeline phases (diagnostic build tools/experiments/wide_stamps.hip ->
libwide_stamps.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd.graph import _ptr, _stream
nu, F, M = (int(v) for v in (sys.argv[1:4] + ["100", "256", "4"][len(sys.argv) - 1:]))
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwide_stamps.so"))
mesh = gwen_amd.geodesic_mesh(nu, reorder="hilbert")
N = mesh.num_nodes
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).cuda(), N)
tr, tl, tv, umax = g.tiles()
x = torch.randn(M, N, F, device="cuda"); w = torch.randn(F, F, device="cuda") / F ** 0.5; b = torch.randn(F, device="cuda")
out = torch.empty(M, N, F, device="cuda")
NW = 8 if F >= 256 else 16
st = torch.zeros(256 * NW * 6, dtype=torch.int64, device="cuda")
L.gwen_wide_set_stamps.argtypes = [C.c_void_p]
assert L.gwen_wide_set_stamps(C.c_void_p(st.data_ptr())) == 0
L.gwen_gcn_wide_layer_f32.argtypes = [C.c_void_p] * 7 + [C.c_int64] * 8 + [C.c_int, C.c_int64, C.c_void_p]
for _ in range(3):
    rc = L.gwen_gcn_wide_layer_f32(_ptr(tr), _ptr(tl), _ptr(tv), _ptr(x), _ptr(w), _ptr(b), _ptr(out), N, N, F, F, F, M,
                                   N * F, N * F, 1, umax, _stream(torch.device("cuda:0")))
    assert rc == 0
torch.cuda.synchronize()
t = st.view(256, NW, 6).double().cpu()
names = ["everything between barrier and the next wait", "vmcnt wait", "barrier", "-", "-", "-"]
tot = t.sum(-1)
print(f"nu={nu} F={F} M={M}: per-wave stamped ticks (100 MHz s_memtime = 10 ns?) mean total {tot.mean():.0f}, min {tot.min():.0f}, max {tot.max():.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:16s} mean {t[..., k].mean():10.0f}  ({100 * t[..., k].mean() / tot.mean():5.1f} %)   wave0 {t[:, 0, k].mean():10.0f}  wave{NW-1} {t[:, NW-1, k].mean():10.0f}")
