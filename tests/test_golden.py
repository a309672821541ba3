"""Golden vectors (tests/golden/c1_gnn_forward.npz, made by tests/golden/make_golden.py).

CPU: the oracle reproduces the committed vectors bit for bit (guards the oracle against drift).
GPU: the HIP path matches the committed per-layer outputs within the stated tolerance.
PARITY UNPINNED at the reference level -- see the generator's docstring.
"""
import os

import numpy as np
import pytest
import torch

from helpers import REL_TOL, rel_err

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_gnn_forward.npz")
LAYERS = [("down_conv_layers.conv1", 1), ("down_conv_layers.conv2", 1), ("down_conv_layers.conv3", 1),
          ("up_conv_layers.upconv3", 1), ("up_conv_layers.upconv4", 1), ("up_conv_layers.upconv5", 0)]


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(PATH))


def test_fixture_shape(gold):
    assert gold["edge_index"].shape == (2, 6000) and gold["x"].shape == (1002, 8)
    assert gold["y/up_conv_layers.upconv5"].shape == (1002, 8)
    assert os.path.getsize(PATH) < 400_000


def test_oracle_reproduces_golden_bitwise(gold):
    from oracle import gcn_oracle as O
    torch.set_num_threads(1)
    ei = torch.from_numpy(gold["edge_index"].astype(np.int64))
    t = torch.from_numpy(gold["x"])
    for name, act in LAYERS:
        w = torch.from_numpy(gold[f"w/conv_layers.{name}.lin.weight"])
        b = torch.from_numpy(gold[f"w/conv_layers.{name}.bias"])
        t = O.gcn_conv(t, ei, w, b)
        t = torch.relu(t) if act else t
        # bitwise only where the contraction has one order: compare with a tight tolerance instead
        assert rel_err(t, gold["y/" + name]) < 1e-6
    _, wn = O.gcn_norm(ei, None, 1002)
    assert wn[:64].numpy().tobytes() == gold["norm_w_head"].tobytes()


def test_c_oracle_matches_golden(gold, cref):
    ei = gold["edge_index"].astype(np.int64)
    t = gold["x"]
    for name, act in LAYERS:
        t = cref.conv(t, ei, gold[f"w/conv_layers.{name}.lin.weight"], gold[f"w/conv_layers.{name}.bias"],
                      relu=bool(act))
        assert rel_err(t, gold["y/" + name]) < 1e-6
    from gwen_amd.mesh import complete_graph
    y = cref.conv(gold["k125/x"], complete_graph(125), gold["k125/w"], gold["k125/b"])
    assert rel_err(y, gold["k125/y"]) < 1e-6
    mean = (gold["k125/x"].astype(np.float64) @ gold["k125/w"].astype(np.float64).T).mean(0) + gold["k125/b"]
    assert rel_err(gold["k125/y"], np.broadcast_to(mean, gold["k125/y"].shape)) < 1e-6   # K_N identity


@pytest.mark.gpu
def test_hip_path_matches_golden(gold, hip_lib):
    import gwen_amd
    dev = "cuda:0"
    ei = torch.from_numpy(gold["edge_index"].astype(np.int64)).to(dev)
    model = gwen_amd.GNNModel(gwen_amd.GNNConfig(1002, 1002, 8, 8, 16))
    sd = model.state_dict()
    for k in sd:
        key = "w/" + k
        if key in gold:
            sd[k] = torch.from_numpy(gold[key])
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    x = torch.from_numpy(gold["x"]).to(dev)
    with torch.no_grad():
        out = model(x, ei)
        assert rel_err(out, gold["y/up_conv_layers.upconv5"]) <= REL_TOL
        # per layer, through the drop-in GCNConv modules
        d, u = model.conv_layers.down_conv_layers, model.conv_layers.up_conv_layers
        t = x
        for (name, act), conv in zip(LAYERS, [d.conv1, d.conv2, d.conv3, u.upconv3, u.upconv4, u.upconv5]):
            t = conv(t, ei, relu=bool(act))
            assert rel_err(t, gold["y/" + name]) <= REL_TOL, name
        k = torch.from_numpy(gwen_amd.complete_graph(125)).to(dev)
        conv = gwen_amd.GCNConv(16, 8).to(dev)
        conv.lin.weight.copy_(torch.from_numpy(gold["k125/w"])); conv.bias.copy_(torch.from_numpy(gold["k125/b"]))
        y = conv(torch.from_numpy(gold["k125/x"]).to(dev), k)
        assert rel_err(y, gold["k125/y"]) <= REL_TOL
