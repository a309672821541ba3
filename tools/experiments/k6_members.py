"""K6 edge kernel on the block-diagonal graph of M members (c5's per-GPU load): us per launch, HBM fraction by
compulsory bytes.  python tools/experiments/k6_members.py [F] [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import ops, interaction as I
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
ei = torch.from_numpy(mesh.edge_index).cuda()
g1 = I.interaction_graph(ei[:, ei[0] != ei[1]], mesh.num_nodes, mesh.num_nodes)
g = g1.batched(M)
net = I.InteractionNet(F).cuda().eval()
n, e = g.num_dst, g.num_edges
x = torch.randn(n, F, device="cuda"); ef = torch.randn(e, F, device="cuda")
with torch.no_grad():
    we, wa, wn, bn = net._weight_blocks()
    p = ops.linear(x, wn, bn, exact=False)
    run = lambda: I.mlp2(ef, we, net.edge_mlp[2].weight, net.edge_mlp[2].bias, g1=p[:, :F], idx1=g.src, g2=p[:, F:2 * F],
                         idx2=g.dst, res=ef, act="silu", graph=g, mean=False, want_out=True)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:                 # steady state: the clocks ramp for tens of ms after idling
        for _ in range(3): run()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
comp = 4 * F * (2 * e + 2 * n) + 12 * e + 4 * n
print(f"F={F} members={M}: E={e} N={n}  {us:.1f} us/launch  {e / us / 1e3:.2f} G edges/s  compulsory {comp / 1e6:.0f} MB -> {comp / us / 1e6:.2f} TB/s = {comp / us / 8e6:.3f} of the HBM peak")
