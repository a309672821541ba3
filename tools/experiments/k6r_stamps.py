"""In-kernel s_memtime stamps of the row-stationary K6 (diagnostic build of interact_rows.hip with -DGWEN_K6R_STAMPS
linked into a private copy of the library).  Slots: 0 step-0 drain  1 other steps' DMA waits  2 step barriers
3 step bodies (DMA issue, loads, MFMA)  4 the activation step's body  5 stores + next G2 rows  6 aggregation."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import _lib
from gwen_amd.interaction import InteractionNet, interaction_graph
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
ei = torch.from_numpy(mesh.edge_index).cuda()
keep = ei[0] != ei[1]
g = interaction_graph(ei[:, keep], mesh.num_nodes, mesh.num_nodes)
net = InteractionNet(F).cuda().eval()
x = torch.randn(mesh.num_nodes, F, device="cuda"); e = torch.randn(g.num_edges, F, device="cuda")
L = _lib.lib()
st = torch.zeros(512 * 8 * 8, dtype=torch.int64, device="cuda")
L.gwen_k6r_set_stamps.argtypes = [C.c_void_p]
from gwen_amd import ops, interaction as I
with torch.no_grad():
    we, wa, wn, bn = net._weight_blocks()
    p = ops.linear(x, wn, bn, exact=False)
    ps, pd = p[:, :F], p[:, F:2 * F]
    run = lambda: I.mlp2(e, we, net.edge_mlp[2].weight, net.edge_mlp[2].bias, g1=ps, idx1=g.src, g2=pd, idx2=g.dst,
                         res=e, act="silu", graph=g, mean=False, want_out=True)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert L.gwen_k6r_set_stamps(C.c_void_p(st.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
t = st.view(512, 8, 8).double().cpu(); t = t[t.sum((1, 2)) > 0]
tot = t.sum(-1)
names = ["step-0 drain", "DMA waits", "step barriers", "step bodies", "activation step", "agg: y tile + barriers", "agg: stores + G2", "agg: sums"]
print(f"per-wave stamped cycles: mean total {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
for k, nm in enumerate(names):
    print(f"  {nm:16s} mean {t[..., k].mean():10.0f} ({100 * t[..., k].mean() / tot.mean():5.1f} %)  wave0 {t[:, 0, k].mean():10.0f} wave7 {t[:, 7, k].mean():10.0f}")
