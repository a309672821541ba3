"""One-off randomized sweep (not a test: tests/ has fixed seeds): the layer kernels the planner can pick -- K8 on f16x3 /
bf16x3 / bf16x6, K4, K3 + K2 -- on random meshes (two node orders), random multigraphs and clustered graphs, random widths,
member counts and row / column scales, each against the fp64 C oracle.   python tools/experiments/stress_layers.py [N_CASES] [SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, gwen_amd
from gwen_amd import ops
from oracle import gcn_ref
from helpers import random_multigraph, rel_err
gcn_ref.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = "cuda:0"
worst = {}
t0 = time.time()
for case in range(ncases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    kind = ["mesh_h", "mesh_m", "multi", "mesh_h"][case % 4]
    if kind.startswith("mesh"):
        nu = int(rng.integers(3, 30))
        m = gwen_amd.geodesic_mesh(nu, reorder="hilbert" if kind == "mesh_h" else "morton")
        n, ei = m.num_nodes, torch.from_numpy(m.edge_index)
    else:
        n = int(rng.integers(2, 3000))
        ei = random_multigraph(n, int(rng.integers(1, 8 * n)), seed=case, self_loops=int(rng.integers(0, 20)),
                               dup=int(rng.integers(0, 50)), isolate=int(rng.integers(0, min(5, n - 1) + 1)))
    fin, fout = int(rng.choice([16, 32, 64, 128, 256])), int(rng.choice([16, 32, 64, 128, 256]))
    members = int(rng.integers(1, 4))
    g = gwen_amd.prepare_graph(ei.to(dev), n)
    gen = torch.Generator().manual_seed(case)
    x = torch.randn(members, n, fin, generator=gen)
    if case % 3 == 0:
        x = x * torch.exp2(torch.randint(-20, 21, (members, n, 1), generator=gen).float())
    w = torch.randn(fout, fin, generator=gen) / fin ** 0.5
    if case % 5 == 0:
        w = w * torch.exp2(torch.randint(-10, 11, (fout, 1), generator=gen).float())
    b = torch.randn(fout, generator=gen) * 0.1
    relu = bool(case % 2)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    refs = [gcn_ref.conv(x[k].numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=relu, f64=True) for k in range(members)]
    ref = torch.from_numpy(np.stack(refs))
    runs = {}
    if ops.wide_supported(fin, fout) and g.tiles() is not None:
        for c in ("f16x3", "3xbf16") + (("bf16x6",) if g.tiles()[3] <= 128 else ()):
            runs["K8/" + c] = lambda c=c: ops.wide_layer(g, xd, wd, bd, relu=relu, contract=c)
    if ops.layer_supported(fin, fout) and n * fin * 4 < 2 ** 32:
        for c in ("bf16x6", "3xbf16", "fp32"):
            runs["K4/" + c] = lambda c=c: ops.layer_fused(g, xd, wd, bd, relu=relu, contract=c)
    runs["K3+K2"] = lambda: ops.propagate(g, ops.linear(xd, wd, None, contract="bf16x6"), bd, relu=relu)
    for name, fn in runs.items():
        got = fn()
        assert bool(torch.isfinite(got).all()), (case, name)
        again = fn()
        assert torch.equal(got, again), (case, name, "not reproducible")
        err = rel_err(got, ref)
        tol = 3e-5 if "3xbf16" in name else 2e-6
        worst[name] = max(worst.get(name, 0.0), err)
        assert err <= tol, (case, kind, n, fin, fout, members, name, err)
print(f"{ncases} cases in {time.time() - t0:.0f} s; worst relative error per path:")
for k, v in sorted(worst.items()):
    print(f"  {k:12s} {v:.2e}")
