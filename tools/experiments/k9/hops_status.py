import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gwen_amd
from gwen_amd import _lib
from gwen_amd.graph import _ptr, _stream
for nu in (6, 14, 30, 60, 100, 150):
    for reorder in ("hilbert", "morton"):
        m = gwen_amd.geodesic_mesh(nu, reorder=reorder)
        g = gwen_amd.prepare_graph(torch.from_numpy(m.edge_index).cuda(), m.num_nodes)
        n = m.num_nodes; t = (n + 63) // 64
        for H in (3, 4):
            h_cnt = torch.zeros(t * 6, dtype=torch.int32, device="cuda"); h_rows = torch.empty(t * 288, dtype=torch.int32, device="cuda")
            h_lid = torch.empty(t * 192 * 8, dtype=torch.int16, device="cuda"); h_val = torch.empty(t * 192 * 8, device="cuda")
            st = torch.empty(3, dtype=torch.int32, device="cuda")
            rc = _lib.lib().gwen_gcn_hops64(_ptr(g.rowptr), _ptr(g.col), _ptr(g.val), n, H, _ptr(h_cnt), _ptr(h_rows), _ptr(h_lid), _ptr(h_val), _ptr(st), _stream(torch.device("cuda:0")))
            print(nu, reorder, "H", H, "rc", rc, "status", st.tolist())
