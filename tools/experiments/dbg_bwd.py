import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, gwen_amd as ga
from oracle import gcn_oracle as O
from helpers import rel_err
DEV="cuda:0"; SEED=23
C,H,members=64,64,3
m = ga.geodesic_mesh(7, reorder="hilbert"); n=m.num_nodes; ei=torch.from_numpy(m.edge_index)
torch.manual_seed(SEED)
ref = O.OracleGNNModel(O.OracleGNNConfig(n, n, C, C, H))
with torch.no_grad():
    for p in ref.parameters():
        if p.dim()==1: p.normal_(0,0.1)
model = ga.GNNModel(ga.GNNConfig(n,n,C,C,H)); model.load_state_dict(ref.state_dict()); model=model.to(DEV)
x = torch.randn(members,n,C,generator=torch.Generator().manual_seed(SEED))
gout = torch.randn(members,n,C,generator=torch.Generator().manual_seed(SEED+1))
xr = x.clone().requires_grad_()
torch.stack([ref(xr[k], ei) for k in range(members)]).backward(gout)
xd = x.to(DEV).requires_grad_()
out = model(xd, ei.to(DEV)); out.backward(gout.to(DEV))
for k in range(members):
    d=(xd.grad[k].cpu().double()-xr.grad[k].double())
    print("member",k,"l2",float(d.norm()/xr.grad[k].double().norm()),"max",float(d.abs().max()), "rows off", int((d.abs().amax(1)>1e-5).sum()))
# per-layer path
xd2 = x.to(DEV).requires_grad_()
g = model.prepare(ei.to(DEV), n)
model.zero_grad()
o2 = model.conv_layers(xd2, g); o2.backward(gout.to(DEV))
print("stack vs per-layer fwd equal", torch.equal(o2, out), "grad rel", rel_err(xd2.grad, xd.grad))
for k in range(members):
    d=(xd2.grad[k].cpu().double()-xr.grad[k].double())
    print("per-layer member",k,"l2",float(d.norm()/xr.grad[k].double().norm()))
# single-member runs through the stack
for k in range(members):
    xs = x[k].to(DEV).requires_grad_(); model.zero_grad()
    model(xs, ei.to(DEV)).backward(gout[k].to(DEV))
    print("single", k, rel_err(xs.grad, xd.grad[k]), float((xs.grad.cpu().double()-xr.grad[k].double()).norm()/xr.grad[k].double().norm()))
