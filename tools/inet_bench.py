"""Time the InteractionNet block (K6) on the c2 mesh: per-launch times, HBM and MFMA figures.
Usage: python tools/inet_bench.py [--channels 64] [--nu 100] [--iters 50]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gwen_amd
from gwen_amd import ops
from gwen_amd.interaction import InteractionNet, interaction_graph, mlp2


def timed(fn, iters):
    import time
    t0 = time.perf_counter()             # 0.1 s of the same work first: the clocks ramp for tens of ms after idling
    while time.perf_counter() - t0 < 0.1:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--nu", type=int, default=100)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--graph", default="mesh", choices=["mesh", "g2m", "m2g"])
    ap.add_argument("--reorder", default="morton", choices=["none", "morton", "hilbert"])
    ap.add_argument("--act", default="silu", choices=["none", "relu", "silu"])
    args = ap.parse_args()
    dev = "cuda:0"
    F = args.channels
    mesh = gwen_amd.geodesic_mesh(args.nu, reorder=None if args.reorder == 'none' else args.reorder)
    if args.graph == "mesh":
        ei, ns, nd = torch.from_numpy(mesh.edge_index), mesh.num_nodes, mesh.num_nodes
    else:
        from gwen_amd import g2m
        a, b = g2m.grid_mesh_edges(mesh)
        ei, ns, nd = (torch.from_numpy(a), mesh.faces.shape[0], mesh.num_nodes) if args.graph == "g2m" \
            else (torch.from_numpy(b), mesh.num_nodes, mesh.faces.shape[0])
    g = interaction_graph(ei.to(dev), ns, nd)
    E = g.num_edges
    torch.manual_seed(23)
    net = InteractionNet(F, args.act).to(dev)
    xs = torch.randn(ns, F, device=dev)
    xd = xs if args.graph == "mesh" else torch.randn(nd, F, device=dev)
    e = torch.randn(E, F, device=dev)
    with torch.no_grad():
        w1, b1 = net.edge_mlp[0].weight, net.edge_mlp[0].bias
        w1e, w1s, w1d = (w1[:, :F].contiguous(), w1[:, F:2 * F].contiguous(), w1[:, 2 * F:].contiguous())
        w2, b2 = net.edge_mlp[2].weight, net.edge_mlp[2].bias
        ps = ops.linear(xs, w1s, None, exact=False)
        pd = ops.linear(xd, w1d, b1, exact=False)
        t_proj = timed(lambda: ops.linear(xs, w1s, None, exact=False), args.iters)
        t_edge = timed(lambda: mlp2(e, w1e, w2, b2, g1=ps, idx1=g.src, g2=pd, idx2=g.dst, res=e, graph=g, act=args.act), args.iters)
        t_edge_noagg = timed(lambda: mlp2(e, w1e, w2, b2, g1=ps, idx1=g.src, g2=pd, idx2=g.dst, res=e, act=args.act), args.iters)
        t_edge_plain = timed(lambda: mlp2(e, w1e, w2, b2, res=e, act=args.act), args.iters)
        _, agg = mlp2(e, w1e, w2, b2, g1=ps, idx1=g.src, g2=pd, idx2=g.dst, res=e, graph=g, act=args.act)
        w3 = net.node_mlp[0].weight
        q = ops.linear(xd, w3[:, :F].contiguous(), net.node_mlp[0].bias, exact=False)
        w3b = w3[:, F:].contiguous()
        t_node = timed(lambda: mlp2(agg, w3b, net.node_mlp[2].weight, net.node_mlp[2].bias, g1=q, res=xd, act=args.act), args.iters)
        t_block = timed(lambda: net(xs, xd, e, g), args.iters)
    # algorithmic bytes of the edge launch: e read + e' write + two gathered rows + agg write + indices
    b_alg = 4 * F * (4 * E + nd) + 8 * E + 4 * nd
    flops = 4 * F * F * E                     # two F x F contractions per edge, fp32-equivalent
    print(f"graph {args.graph}: Ns={ns} Nd={nd} E={E} F={F} max_degree={g.max_degree}")
    print(f"node projection (K3)        {t_proj:8.1f} us")
    print(f"edge MLP + aggregate (K6)   {t_edge:8.1f} us   {b_alg / t_edge / 1e3:7.1f} GB/s algorithmic, "
          f"{flops / t_edge / 1e6:6.1f} TFLOP/s fp32-equivalent ({3 * flops / t_edge / 1e6:6.1f} bf16 issued), "
          f"{E / t_edge / 1e3:6.2f} G edges/s")
    print(f"  without aggregation       {t_edge_noagg:8.1f} us")
    print(f"  without gathers either    {t_edge_plain:8.1f} us")
    print(f"node MLP (K6)               {t_node:8.1f} us")
    print(f"whole block (1 K3 + 2 K6)    {t_block:8.1f} us   {E / t_block / 1e3:6.2f} G edges/s")


if __name__ == "__main__":
    main()
