"""End-to-end timing of the reference's OWN evaluation loop shape (models_gnn.py:428-465) through
gwen_amd.data.eval_loop: members = graph nodes of a complete graph, features = flattened fields, one time
index = ceil(N / batch_size) full-graph batches (host -> device copy, forward, masked L1 loss, un-permute).
    python tools/refloop.py [members] [height] [ncells] [hidden] [batch_size] [time_indices]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gwen_amd
from gwen_amd import data
a = [int(v) for v in sys.argv[1:]]
n, hgt, cells, hid, bs, T = (a + [125, 16, 1024, 1024, 1, 3][len(a):])[:6]
dev = "cuda:0"
rng = np.random.default_rng(23)
ds = data.MemberGraphDataset(rng.standard_normal((T, n, hgt, cells), dtype=np.float32), split=n // 2, seed=23)
c = ds.channels
torch.manual_seed(23)
model = gwen_amd.GNNModel(gwen_amd.GNNConfig(n, n, c, c, hid))
data.eval_loop(model, data.MemberGraphDataset(ds.data[:1], n // 2, 23), bs, dev)        # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
loss, outs = data.eval_loop(model, ds, bs, dev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nb = len(outs)
print(f"members={n} channels={c} hidden={hid} batch_size={bs}: {T} time indices, {nb} batches, "
      f"{dt / T * 1e3:.1f} ms per time index, {dt / nb * 1e6:.0f} us per batch (forward alone ~0.13 ms)")
