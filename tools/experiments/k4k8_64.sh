for m in 1 2 4; do for c in 3xbf16 bf16x6 f16x3; do python tools/experiments/k8_one.py 64 64 $m $c; done; done
python - <<'PY'
import sys; sys.path.insert(0,'.')
import torch, gwen_amd
from gwen_amd import ops
dev="cuda:0"
mesh=gwen_amd.geodesic_mesh(100,reorder="hilbert")
g=gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev),mesh.num_nodes)
for M in (1,2,4):
    x=torch.randn(M,mesh.num_nodes,64,device=dev); w=torch.randn(64,64,device=dev)/8; b=torch.randn(64,device=dev)
    for c in ("3xbf16","bf16x6"):
        f=lambda: ops.layer_fused(g,x,w,b,relu=True,contract=c)
        for _ in range(30): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        print(f"K4 64->64 x {M} {c}: {e0.elapsed_time(e1)/50*1e3:.1f} us")
    # K2 alone
    f=lambda: ops.propagate(g,x,None,relu=False)
    for _ in range(30): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"K2 64 x {M}: {e0.elapsed_time(e1)/50*1e3:.1f} us")
PY
