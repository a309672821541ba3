import sys; sys.path.insert(0, "/root/repo")
import torch
from gwen_amd import ops
dev="cuda:0"
for rows, f in ((600000, 64), (100002, 64), (200000, 64), (100002, 256), (200000, 256), (600000, 256), (100002, 128)):
    g = torch.randn(rows, f, device=dev); x = torch.randn(rows, f, device=dev)
    want = g.double().t() @ x.double()
    line = f"grad_weight rows={rows} F={f}:"
    for c in ("fp32", "bf16x6", "3xbf16"):
        got = ops.grad_weight(g, x, c)
        err = float((got.double()-want).abs().max()/want.abs().max())
        for _ in range(3): ops.grad_weight(g, x, c)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): ops.grad_weight(g, x, c)
        b.record(); torch.cuda.synchronize()
        line += f"  {c} {a.elapsed_time(b) * 50:.1f} us ({err:.1e})"
    print(line)
