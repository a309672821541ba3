"""In-kernel s_memtime stamps of K9's phases (diagnostic build: hops.hip with -DGWEN_HOPS_STAMPS -> libhops_stamps.so).
slots: 0 staging (wait for prefetched rows, LDS writes)  1 barrier  2 gathers  3 barriers after them
       4+2s epilogue of stage s  5+2s barrier after it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd.graph import _ptr, _stream
nu, M, S = (int(v) for v in (sys.argv[1:4] + ["100", "1", "3"][len(sys.argv) - 1:]))
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhops_stamps.so"))
mesh = gwen_amd.geodesic_mesh(nu, reorder="hilbert")
N = mesh.num_nodes
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).cuda(), N)
h_cnt, h_rows, h_lid, h_val, hh = g.hops(3)
x = torch.randn(M, N, 32, device="cuda"); out = torch.empty(M, N, 32, device="cuda")
Ws = [None, torch.randn(16, 32, device="cuda") * .2, torch.randn(32, 16, device="cuda") * .2][:S]
bs = [torch.randn(32, device="cuda"), torch.randn(16, device="cuda"), torch.randn(32, device="cuda")][:S]
fin, fout = [32, 32, 16][:S], [32, 16, 32][:S]
nb = min(((N + 63) // 64) * M, 512)
st = torch.zeros(nb * 8 * 12, dtype=torch.int64, device="cuda")
L.gwen_hops_set_stamps.argtypes = [C.c_void_p]
assert L.gwen_hops_set_stamps(C.c_void_p(st.data_ptr())) == 0
PP = C.c_void_p * S; II = C.c_int32 * S
fn = L.gwen_gcn_narrow_chain_f32
fn.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, PP, PP, II, II, II, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
for _ in range(3):
    rc = fn(_ptr(h_cnt), _ptr(h_rows), _ptr(h_lid), _ptr(h_val), hh, _ptr(x), _ptr(out), N, S,
            PP(*[None if w is None else w.data_ptr() for w in Ws]), PP(*[b.data_ptr() for b in bs]), II(*fin), II(*fout), II(*([1] * S)),
            M, N * 32, N * fout[-1], _stream(torch.device("cuda:0")))
    assert rc == 0
torch.cuda.synchronize()
t = st.view(nb, 8, 12).double().cpu()
tot = t.sum(-1)
print(f"nu={nu} M={M} S={S}: per-wave stamped ticks mean total {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}; tiles per block {((N + 63) // 64) * M / nb:.2f}")
for k in range(4 + 2 * S):
    print(f"  slot {k:2d} mean {t[..., k].mean():9.0f} ({100 * t[..., k].mean() / tot.mean():5.1f} %)  wave0 {t[:, 0, k].mean():9.0f} wave7 {t[:, 7, k].mean():9.0f}")
