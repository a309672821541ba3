"""EXPERIMENT: the row-streaming 64-channel dense linear (rows64.hip) against ops.linear (K3 on K8's pipeline)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from gwen_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "librows64.so"))
dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
def t(fn, k=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / k * 1e3
for rows in (100002, 200000, 600000):
    for fout in (64, 128, 192):
        torch.manual_seed(rows + fout)
        x = torch.randn(rows, 64, device=dev); w = torch.randn(fout, 64, device=dev) / 8; b = torch.randn(fout, device=dev)
        out = torch.empty(rows, fout, device=dev)
        want = x.double() @ w.double().t() + b.double()
        line = f"rows {rows} 64 -> {fout}:"
        for ns, name in ((2, "3xbf16"), (3, "bf16x6")):
            best = None
            for blocks in (512, 1024, 2048):
                run = lambda: lib.rows64_launch(P(x), P(w), P(b), P(out), rows, fout, ns, 0, blocks, st)
                assert run() == 0
                torch.cuda.synchronize()
                err = float((out.double() - want).abs().max() / want.abs().max())
                us = t(run)
                best = (us, blocks, err) if best is None or us < best[0] else best
            ref = ops.linear(x, w, b, contract=name)
            same = bool(torch.equal(ref, out)) if best[1] == 2048 else None
            us_ref = t(lambda: ops.linear(x, w, b, contract=name))
            line += f"  {name}: rows64 {best[0]:.1f} us ({best[1]} blocks, err {best[2]:.1e}) vs ops.linear {us_ref:.1f}"
        print(line)
