"""ops.grad_weight on every contraction against fp64, and its timing at the InteractionNet's edge-level shape:
    python tools/experiments/gradw_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gwen_amd import ops
dev = "cuda:0"
torch.manual_seed(3)
for rows, fout, fin in ((5000, 128, 128), (70001, 256, 256), (3000, 256, 64), (2047, 64, 256), (2049, 128, 256), (1, 256, 256)):
    g = torch.randn(rows, fout, device=dev) * torch.exp2(torch.randint(-6, 7, (rows, 1), device=dev).float())
    x = torch.randn(rows, fin, device=dev)
    want = g.double().t() @ x.double()
    for c in ("fp32", "bf16x6", "3xbf16", "f16x3"):
        got = ops.grad_weight(g, x, c)
        err = float((got.double() - want).abs().max() / want.abs().max())
        again = ops.grad_weight(g, x, c)
        print(f"rows={rows} {fout}x{fin} {c:7s} rel err {err:.2e} bitwise-repeat {bool(torch.equal(got, again))}")
for rows, f in ((600000, 256), (100002, 256), (600000, 128), (600000, 64)):
    g = torch.randn(rows, f, device=dev); x = torch.randn(rows, f, device=dev)
    line = f"grad_weight rows={rows} F={f}:"
    for c in ("fp32", "bf16x6", "3xbf16"):
        for _ in range(3):
            ops.grad_weight(g, x, c)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.grad_weight(g, x, c)
        b.record(); torch.cuda.synchronize()
        line += f"  {c} {a.elapsed_time(b) * 100:.1f} us"
    print(line)
