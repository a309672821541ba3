"""K3 (3xbf16) timing: the 128x128-tile kernel (GWEN_K3_DENSE=0) against the K8 pipeline without a graph (=1).
python tools/experiments/k3_time.py  -> lines 'rows fin fout us GB/s err'"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import ops
dev = "cuda:0"
torch.manual_seed(5)
def timed(fn, k=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
print("GWEN_K3_DENSE =", os.environ.get("GWEN_K3_DENSE"))
for rows, fin, fout in [(100002, 256, 256), (400008, 256, 256), (400008, 256, 768), (100002, 128, 128), (100002, 64, 64),
                        (100002, 64, 256), (200000, 128, 256), (32768, 256, 256), (8192, 256, 256), (100002, 256, 64)]:
    x = torch.randn(rows, fin, device=dev); w = torch.randn(fout, fin, device=dev) / fin ** .5; b = torch.randn(fout, device=dev)
    out = ops.linear(x, w, b, relu=True, exact=False)
    ref = torch.relu(x[:4096].double() @ w.double().t() + b.double())
    err = ((out[:4096].double() - ref).abs().max() / ref.abs().max()).item()
    tail = torch.relu(x[-70:].double() @ w.double().t() + b.double())
    err2 = ((out[-70:].double() - tail).abs().max() / tail.abs().max()).item()
    us = timed(lambda: ops.linear(x, w, b, relu=True, exact=False))
    print(f"{rows:7d} {fin:4d} {fout:4d}  {us:8.1f} us  {rows * (fin + fout) * 4 / us / 1e6:6.2f} TB/s  err {err:.1e} tail {err2:.1e}  sum {out.double().sum().item():.10e}")
