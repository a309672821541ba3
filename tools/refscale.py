"""Timing of the reference's OWN workload shape on the device: complete graph over ~125 ensemble members,
flattened fields as features, hidden_feats 1024 (/root/reference/src/gwen/config.json:9,12) -- per-kernel
hipEvent times through the stack launcher, on each precision.   python tools/refscale.py [members] [channels] [hidden]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gwen_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125
c = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = "cuda:0"
ei = torch.from_numpy(gwen_amd.complete_graph(n)).to(dev)
torch.manual_seed(23)
model = gwen_amd.GNNModel(gwen_amd.GNNConfig(n, n, c, c, h)).to(dev).eval()
x = torch.randn(n, c, device=dev)
g = model.prepare(ei, n)
import time
for prec in ("bf16x6", "3xbf16", "fp32"):
    model.set_precision(prec)
    plan = gwen_amd.StackForward(model.stack(), g, model._packed_weights(g))
    ev = gwen_amd.KernelEvents(12)
    for _ in range(5):
        plan.run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        out = plan.run(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    plan.run(x, events=ev)
    print(f"N={n} C={c} H={h} precision {prec}: {dt*1e6:.1f} us per forward")
    for kind, layer, fin, fout, sec in ev.durations():
        print(f"   layer {layer} {kind:9s} {fin:6d} -> {fout:6d}  {sec*1e6:8.1f} us")
