"""In-kernel s_memtime stamps of K8's step (diagnostic build: python -m gwen_amd.build --variant stamp --src wide.hip -DK8_STAMP=1):
   GWEN_HIP_LIB=$PWD/gwen_amd/variants/libgwen_hip.stamp.so python tools/experiments/k8_stamp.py [F] [M]
One block's waves stamp five points of 32 consecutive steps: before the vmcnt wait (0), after it (1), after the barrier (2),
after the first half of the regions (3), after the last region (4).  Printed: mean cycles per phase and wave, and the skew."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gwen_amd
from gwen_amd import _lib
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
mesh = gwen_amd.geodesic_mesh(100, reorder="hilbert")
n = mesh.num_nodes
torch.manual_seed(5)
w = torch.randn(F, F, device=dev) / F ** 0.5
b = torch.randn(F, device=dev)
g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), n)
plan = gwen_amd.StackForward([(w, b, True, "auto_x3")] * 4, g)
x = torch.randn(M, n, F, device=dev)
for _ in range(20):
    out = plan.run(x)
torch.cuda.synchronize()
L = _lib.lib()
NWV = 8 if F >= 256 else 16
buf = np.zeros(16 * 32 * 8, dtype=np.uint32)
assert L.gwen_k8_stamps_read(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(16, 32, 8)[:NWV, :, :5].astype(np.int64)
# s_memtime ticks: convert with the step count known per tile; report raw ticks
names = ["vmcnt wait", "barrier", "regions first half", "regions second half", "end -> next step's wait (top-of-step code)"]
d = np.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3]], -1) & 0xffffffff
nxt = (t[:, 1:, 0] - t[:, :-1, 4]) & 0xffffffff
step = (t[:, 1:, 0] - t[:, :-1, 0]) & 0xffffffff
print(f"F={F} M={M}: step = {step.mean():.0f} ticks (min {step.min()}, max {step.max()})")
for k in range(4):
    print(f"  {names[k]:44s} mean {d[:, :, k].mean():7.0f}  per wave " + " ".join(f"{v:6.0f}" for v in d[:, :, k].mean(1)))
print(f"  {names[4]:44s} mean {nxt.mean():7.0f}  per wave " + " ".join(f"{v:6.0f}" for v in nxt.mean(1)))
nc = F // 64
for c in range(nc):
    sel = d[:, c::nc, :]
    print(f"  chunk {c} of a tile: wait {sel[:, :, 0].mean():6.0f} barrier {sel[:, :, 1].mean():6.0f} first {sel[:, :, 2].mean():6.0f} second {sel[:, :, 3].mean():6.0f}")
arr = t[:, :, 0]
print("  skew of arrival at the wait (max - min over waves), mean over steps:", float((arr.max(0) - arr.min(0)).mean()))
