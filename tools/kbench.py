"""Per-kernel timing on the GPU box (HIP events on torch's current stream, which is the stream the
C-ABI launchers are given).  Usage: python tools/kbench.py [F ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gwen_amd
from gwen_amd import ops

def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts)//2], ts[len(ts)//10], ts[(9*len(ts))//10]

def main():
    Fs = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 256]
    for reorder in (None, "morton"):
        mesh = gwen_amd.geodesic_mesh(100, reorder=reorder)
        N, E = mesh.num_nodes, mesh.num_edges
        ei = torch.from_numpy(mesh.edge_index).cuda()
        t0 = time.time(); g = gwen_amd.prepare_graph(ei, N); torch.cuda.synchronize(); t1 = time.time()
        med, lo, hi = timeit(lambda: gwen_amd.prepare_graph(ei, N, validate=False), iters=20, warm=3)
        print(f"reorder={reorder} N={N} E={E} prep first {1e3*(t1-t0):.1f} ms, steady {med:.0f} us")
        for F in Fs:
            h = torch.randn(N, F, device="cuda"); b = torch.randn(F, device="cuda")
            w = torch.randn(F, F, device="cuda") / F ** 0.5
            balg = 4 * F * (E + 2 * N) + 8 * E + 8 * N
            med, lo, hi = timeit(lambda: ops.propagate(g, h, b, True))
            print(f"  F={F:4d} K2 propagate {med:7.1f} us (p10 {lo:.1f} p90 {hi:.1f})  {balg/med/1e6:6.2f} TB/s alg  {E/med/1e3:6.2f} Gedge/s")
            med, lo, hi = timeit(lambda: ops.linear(h, w))
            fl = 2 * N * F * F
            print(f"         K3 linear    {med:7.1f} us  {fl/med/1e6:6.2f} TFLOP/s  {(8*N*F)/med/1e6:5.2f} TB/s")
            med, lo, hi = timeit(lambda: torch.mm(h, w.t()))
            print(f"         torch.mm     {med:7.1f} us  {fl/med/1e6:6.2f} TFLOP/s")
            med, lo, hi = timeit(lambda: h.clone())
            print(f"         copy         {med:7.1f} us  {(8*N*F)/med/1e6:5.2f} TB/s")
if __name__ == "__main__":
    main()
