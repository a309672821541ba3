"""One-off randomized sweep of the InteractionNet block's backward (K6^T) against torch autograd on the fp64 oracle: random
bipartite / same-node graphs with skewed target degrees (empty targets, heavy ones), ragged edge counts, widths 32 .. 256,
every activation, sum / mean, with and without the edge update; every gradient at 1e-4, two runs bitwise equal.
    python tools/experiments/stress_inet_bwd.py [N_CASES] [SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gwen_amd.interaction import InteractionNet, interaction_graph
from oracle import interaction_oracle as IO
from helpers import rel_err
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
DEV = "cuda:0"
worst = 0.0
t0 = time.time()
for case in range(ncases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    F = int(rng.choice([32, 64, 64, 128, 256, 256]))
    act = ["silu", "relu", "none"][case % 3]
    aggr = ["sum", "mean"][(case // 3) % 2]
    bip = bool(case % 2)
    upd = bool((case // 2) % 4 != 0)                     # a quarter of the cases: no edge update (encoder / decoder blocks)
    ns = int(rng.integers(2, 500))
    nd = int(rng.integers(2, 500)) if bip else ns
    e_ = int(rng.integers(1, 5000))
    p = rng.random(nd) ** (4 if case % 3 == 0 else 1) + 1e-9
    dst = rng.choice(nd, size=e_, p=p / p.sum())
    src = rng.integers(0, ns, size=e_)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    torch.manual_seed(case)
    net = InteractionNet(F, act, aggr)
    if act == "relu":          # keep every hidden unit's pre-activation away from the kink (half the units on, half off): a
        with torch.no_grad():  # unit the 17-bit forward and the fp64 oracle put on different sides of 0 flips a whole gradient element
            for lin in (net.edge_mlp[0], net.node_mlp[0]):
                lin.bias.copy_(torch.where(torch.arange(F) % 2 == 0, 200.0, -200.0))
    g = torch.Generator().manual_seed(case)
    xs, xd, ef = torch.randn(ns, F, generator=g), torch.randn(nd, F, generator=g), torch.randn(e_, F, generator=g)
    gxo, geo = torch.randn(nd, F, generator=g), torch.randn(e_, F, generator=g)
    sd = {k: v.double().clone().requires_grad_() for k, v in net.state_dict().items()}
    xs64, xd64, ef64 = xs.double().requires_grad_(), xd.double().requires_grad_(), ef.double().requires_grad_()
    wx, we = IO.interaction(xs64 if bip else xd64, xd64, ef64, ei, sd, act, aggr)
    loss = (wx * gxo.double()).sum()
    if upd:
        loss = loss + (we * geo.double()).sum()
    loss.backward()
    graph = interaction_graph(ei.to(DEV), ns, nd)
    net = net.to(DEV)
    grads = []
    for rep in range(2):
        xsd, xdd = xs.to(DEV).requires_grad_(), xd.to(DEV).requires_grad_()
        efd = graph.sort_edges(ef.to(DEV)).detach().requires_grad_()
        net.zero_grad(set_to_none=True)
        gx, ge = net(xsd if bip else xdd, xdd, efd, graph, update_edges=upd)
        l = (gx * gxo.to(DEV)).sum()
        if upd:
            l = l + (ge * graph.sort_edges(geo.to(DEV))).sum()
        l.backward()
        grads.append([t.grad.clone() for t in ([xdd, efd] + ([xsd] if bip else []) + list(net.parameters()))
                      if t.grad is not None])
    assert len(grads[0]) == len(grads[1]) and all(torch.equal(a, b) for a, b in zip(*grads)), (case, "not reproducible")
    errs = [rel_err(xdd.grad, xd64.grad), rel_err(graph.unsort_edges(efd.grad), ef64.grad)]
    if bip:
        errs.append(rel_err(xsd.grad, xs64.grad))
    for k, pp in net.named_parameters():
        if sd[k].grad is not None and pp.grad is not None:
            errs.append(rel_err(pp.grad, sd[k].grad))
    worst = max(worst, max(errs))
    assert max(errs) <= 1e-4, (case, F, act, aggr, bip, upd, ns, nd, e_, errs)
print(f"{ncases} cases in {time.time() - t0:.0f} s; worst relative gradient error {worst:.2e}")
