"""K4 on the f16x3 split straight through the C ABI (exact = 3) against the plain-C oracle in fp64, and its timing
beside the bf16 splits:   python tools/experiments/k4_f16_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch, gwen_amd
from gwen_amd import _lib
from gwen_amd.graph import _ptr, _stream
from oracle import gcn_ref
gcn_ref.build()
dev = torch.device("cuda:0")
L = _lib.lib()


def run(g, x, w, b, relu, exact, out=None):
    gr, gc, gv = g.grouped()
    n, fin = x.shape
    fout = w.shape[0]
    out = torch.empty(n, fout, device=dev) if out is None else out
    rc = L.gwen_gcn_layer_f32(_ptr(gr), _ptr(gc), _ptr(gv), _ptr(x), _ptr(w), _ptr(b), _ptr(out), n, fin, fout, fin, fout,
                              1, n * fin, n * fout, int(relu), exact, _stream(dev))
    _lib.check(rc, "layer")
    return out


mesh = gwen_amd.geodesic_mesh(13, reorder="hilbert")
ei = torch.from_numpy(mesh.edge_index)
g = gwen_amd.prepare_graph(ei.to(dev), mesh.num_nodes)
worst = 0.0
for fin in (16, 32, 64, 128, 256):
    for fout in (16, 32, 64, 128, 256):
        torch.manual_seed(fin * 1000 + fout)
        x = torch.randn(mesh.num_nodes, fin) * torch.exp2(torch.randint(-20, 21, (mesh.num_nodes, 1)).float())
        w = torch.randn(fout, fin) / fin ** 0.5 * torch.exp2(torch.randint(-8, 9, (fout, 1)).float())
        b = torch.randn(fout) * 0.1
        got = run(g, x.to(dev), w.to(dev), b.to(dev), True, 3).cpu().numpy()
        ref = gcn_ref.conv(x.numpy(), ei.numpy(), w.numpy(), b.numpy(), relu=True, f64=True)
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        worst = max(worst, err)
        if err > 2e-6:
            print("FAIL", fin, fout, err)
print("worst rel err", worst)
# timing at c2 size
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), mesh.num_nodes)
for fin, fout in ((64, 64), (32, 64), (16, 32), (64, 32)):
    x = torch.randn(mesh.num_nodes, fin, device=dev); w = torch.randn(fout, fin, device=dev) / fin ** 0.5
    b = torch.randn(fout, device=dev); out = torch.empty(mesh.num_nodes, fout, device=dev)
    line = f"K4 {fin}->{fout}:"
    for name, ex in (("bf16x3", 0), ("bf16x6", 2), ("f16x3", 3)):
        for _ in range(50):
            run(g, x, w, b, True, ex, out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(200):
                run(g, x, w, b, True, ex, out)
            b_.record(); torch.cuda.synchronize()
            best = min(best, a_.elapsed_time(b_) * 1e3 / 200)
        line += f"  {name} {best:6.2f} us"
    print(line)
