#!/usr/bin/env python3
"""Generates tests/golden/c1_gnn_forward.npz -- the committed golden vectors for BASELINE config c1.

Inputs + weights + per-layer outputs of the CPU oracle (oracle/gcn_oracle.py) on the 1 002-node /
6 000-edge geodesic mesh, seed 23 (the reference's seed, /root/reference/src/gwen/config.json:14).
The reference itself cannot produce these (torch-geometric 2.3.1 is not installable here, SURVEY
8c), so the vectors pin the ORACLE ("parity unpinned" at the reference level): they guard against
silent drift of the oracle and give the GPU tests fixed expected values.

    python tests/golden/make_golden.py          # rewrites the .npz next to this file
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from gwen_amd.mesh import complete_graph, geodesic_mesh   # noqa: E402
from oracle import gcn_oracle as O                          # noqa: E402

SEED, C, H = 23, 8, 16


def main():
    torch.set_num_threads(1)
    torch.manual_seed(SEED)
    mesh = geodesic_mesh(10)
    ei = torch.from_numpy(mesh.edge_index)
    model = O.OracleGNNModel(O.OracleGNNConfig(mesh.num_nodes, mesh.num_nodes, C, C, H))
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.1)
    x = torch.randn(mesh.num_nodes, C)
    out = {"edge_index": mesh.edge_index.astype(np.int32), "x": x.numpy()}
    sd = model.state_dict()
    for k, v in sd.items():
        if v.numel():
            out["w/" + k] = v.numpy()
    t = x
    with torch.no_grad():
        for name, act in (("down_conv_layers.conv1", 1), ("down_conv_layers.conv2", 1),
                          ("down_conv_layers.conv3", 1), ("up_conv_layers.upconv3", 1),
                          ("up_conv_layers.upconv4", 1), ("up_conv_layers.upconv5", 0)):
            t = O.gcn_conv(t, ei, sd[f"conv_layers.{name}.lin.weight"], sd[f"conv_layers.{name}.bias"])
            if act:
                t = torch.relu(t)
            out["y/" + name] = t.numpy()
        assert torch.equal(t, model(x, ei))
        # normalised weights of the mesh (first 64) and of K_7 (all 1/7)
        _, w = O.gcn_norm(ei, None, mesh.num_nodes)
        out["norm_w_head"] = w[:64].numpy()
        # the reference's own graph family: K_125 with 16 channels, one layer
        k125 = torch.from_numpy(complete_graph(125))
        xk = torch.randn(125, 16)
        wk, bk = torch.randn(8, 16) * 0.25, torch.randn(8) * 0.1
        out["k125/x"], out["k125/w"], out["k125/b"] = xk.numpy(), wk.numpy(), bk.numpy()
        out["k125/y"] = O.gcn_conv(xk, k125, wk, bk).numpy()
    path = os.path.join(HERE, "c1_gnn_forward.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", len(out), "arrays")


if __name__ == "__main__":
    main()
