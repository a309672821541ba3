"""Long rows beyond K7's 256-node graphs: K2 on the segment chain (edge-parallel) against the plain row walk
(8 entries per round trip), K_1000 and a 50 000-in-edge star.   python tools/longrow_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
from gwen_amd import _lib, ops
from gwen_amd.graph import _ptr, _stream
dev = torch.device("cuda:0")


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for name, n, ei in (("K_1000", 1000, torch.from_numpy(gwen_amd.complete_graph(1000))),
                    ("star 50k", 50001, torch.stack([torch.arange(1, 50001), torch.zeros(50000, dtype=torch.long)]))):
    g = gwen_amd.prepare_graph(ei.to(dev), n)
    levels = g.long_row_levels()
    for f in (64, 256):
        h = torch.randn(n, f, device=dev)
        out = torch.empty(n, f, device=dev)
        L = _lib.lib()
        plain = timed(lambda: L.gwen_gcn_propagate_f32(_ptr(g.rowptr), _ptr(g.col), _ptr(g.val), _ptr(h), None, _ptr(out),
                                                       n, f, f, f, 1, n * f, n * f, 0, _stream(dev)))
        chain = timed(lambda: ops.propagate(g, h))
        print(f"{name:9s} F={f:3d}: plain row walk {plain:9.1f} us   segment chain ({len(levels)} launches) {chain:8.1f} us"
              f"   x{plain / chain:.1f}")
