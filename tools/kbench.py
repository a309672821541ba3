"""Per-kernel timing on the GPU box, straight through the C ABI (no torch op in the loop).
Average over R back-to-back launches between two hipEvents on the launch stream.
Usage: python tools/kbench.py [k2|k3|k4|all] [F ...]       (tuning knobs via GWEN_* env vars)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gwen_amd
from gwen_amd import _lib
from gwen_amd.graph import _ptr, _stream


def timed(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    Fs = [int(a) for a in sys.argv[2:]] or [16, 32, 64, 128]
    reorder = os.environ.get("KB_REORDER", "morton")
    mesh = gwen_amd.geodesic_mesh(100, reorder=None if reorder == "none" else reorder)
    N, E = mesh.num_nodes, mesh.num_edges
    ei = torch.from_numpy(mesh.edge_index).cuda()
    g = gwen_amd.prepare_graph(ei, N)
    gr, gc, gv = g.grouped()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _stream(dev)
    knobs = {k: v for k, v in os.environ.items() if k.startswith("GWEN_")}
    print(f"N={N} E={E} reorder={reorder} knobs={knobs}")
    for F in Fs:
        h = torch.randn(N, F, device=dev); b = torch.randn(F, device=dev)
        w = torch.randn(F, F, device=dev) / F ** 0.5
        out = torch.empty(N, F, device=dev)
        balg = 4 * F * (E + 2 * N) + 8 * E + 8 * N
        if which in ("k2", "all"):
            t = timed(lambda: L.gwen_gcn_propagate_f32(_ptr(g.rowptr), _ptr(g.col), _ptr(g.val), _ptr(h), _ptr(b), _ptr(out), N, F, F, F, 1, N * F, N * F, 1, st))
            print(f"F={F:4d} K2 propagate {t:7.2f} us  {balg/t/1e6:6.2f} TB/s alg  {E/t/1e3:6.2f} Gedge/s")
        if which in ("k3", "all"):
            t = timed(lambda: L.gwen_gcn_linear_f32(_ptr(h), _ptr(w), None, _ptr(out), N, F, F, F, F, 0, int(os.environ.get('KB_EXACT', '0')), None, 0, st))
            print(f"F={F:4d} K3 linear    {t:7.2f} us  {2*N*F*F/t/1e6:6.2f} TFLOP/s  {(8*N*F)/t/1e6:5.2f} TB/s")
        if which in ("k4", "all") and L.gwen_gcn_layer_supported(F, F):
            t = timed(lambda: L.gwen_gcn_layer_f32(_ptr(gr), _ptr(gc), _ptr(gv), _ptr(h), _ptr(w), _ptr(b), _ptr(out), N, F, F, F, F, 1, N * F, N * F, 1, int(os.environ.get('KB_EXACT', '0')), st))
            print(f"F={F:4d} K4 layer     {t:7.2f} us  {balg/t/1e6:6.2f} TB/s alg  {E/t/1e3:6.2f} Gedge/s")


if __name__ == "__main__":
    main()
