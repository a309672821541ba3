"""Caller-side pieces: dataset, loader-equivalent batching, loops, checkpoint shim."""
import numpy as np
import pytest
import torch

import gwen_amd
from gwen_amd import data as D
from helpers import REL_TOL, SEED, rel_err
from oracle import gcn_oracle as O


def make_dataset(times=3, members=9, height=2, ncells=5, split=6):
    rng = np.random.default_rng(SEED)
    arr = rng.standard_normal((times, members, height, ncells)).astype(np.float32)
    return D.MemberGraphDataset(arr, split=split, seed=SEED), arr


def test_dataset_matches_reference_producer_semantics():
    ds, arr = make_dataset()
    assert len(ds) == 3 and ds.nodes == 9 and ds.channels == 10
    assert ds.edge_index.shape == (2, 72)                       # K_9: utils.py:176
    s = ds.get(1)
    assert s.x.shape == (9, 10) and s.x.dtype == torch.float32
    assert np.array_equal(s.x.numpy(), arr[1].reshape(9, 10))   # stack(features=[height, ncells])
    assert int(s.target_mask.sum()) == 3 and len(ds.input_indices) == 6
    assert set(ds.input_indices) | set(ds.target_indices) == set(range(9))
    assert s.edge_index is ds.edge_index                        # one tensor object for every sample
    with pytest.raises(ValueError):
        D.MemberGraphDataset(np.zeros((2, 3, 4)), 1)


def test_full_graph_batches_are_seed_first_permutations():
    ds, _ = make_dataset()
    s = ds.get(0)
    batches = list(D.full_graph_batches(s, 4))
    assert len(batches) == 3                                    # ceil(9 / 4)
    seen = []
    for b, perm in batches:
        assert sorted(perm.tolist()) == list(range(9))
        assert torch.equal(b.x, s.x[torch.from_numpy(perm)])
        assert b.edge_index is s.edge_index
        seen += perm[: min(4, 9 - len(seen))].tolist()
    assert seen == list(range(9))                               # seeds sweep the nodes in order
    assert D.batch_permutation(6, 2, 2).tolist() == [2, 3, 0, 1, 4, 5]


def test_oracle_is_relabelling_invariant_on_complete_graph():
    # the premise of re-using one prepared graph for every batch
    ds, _ = make_dataset()
    s = ds.get(0)
    torch.manual_seed(SEED)
    m = O.OracleGNNModel(O.OracleGNNConfig(9, 9, 10, 10, 16))
    full = m(s.x, s.edge_index)
    for b, perm in D.full_graph_batches(s, 4):
        assert rel_err(m(b.x, b.edge_index), full[torch.from_numpy(perm)]) < 1e-5


def test_checkpoint_shim():
    torch.manual_seed(SEED)
    ref = O.OracleGNNModel(O.OracleGNNConfig(5, 5, 12, 7, 32))
    sd = D.extract_state_dict(ref)
    assert list(sd) == D.REFERENCE_KEYS and len(sd) == 20
    cfg = D.config_from_state_dict(sd)
    assert (cfg.channels_in, cfg.channels_out, cfg.hidden_feats) == (12, 7, 32)
    model = gwen_amd.GNNModel(cfg)
    model.load_state_dict(sd, strict=True)
    ddp_style = {"module." + k: v for k, v in ref.state_dict().items()}
    assert all(torch.equal(a, b) for a, b in zip(D.extract_state_dict(ddp_style).values(), sd.values()))
    with pytest.raises(KeyError):
        D.extract_state_dict({"conv_layers.down_conv_layers.conv1.bias": torch.zeros(1)})


@pytest.mark.gpu
def test_eval_and_train_loops_on_device(hip_lib):
    ds, _ = make_dataset(times=2, members=125, height=2, ncells=8, split=100)   # reference scale: K_125
    torch.manual_seed(SEED)
    ref = O.OracleGNNModel(O.OracleGNNConfig(125, 125, 16, 16, 32))
    model = gwen_amd.GNNModel(gwen_amd.GNNConfig(125, 125, 16, 16, 32))
    model.load_state_dict(D.extract_state_dict(ref), strict=True)
    cache = gwen_amd.default_cache()
    misses0 = cache.misses
    loss, outs = D.eval_loop(model, ds, batch_size=50, device="cuda:0")
    assert cache.misses - misses0 == 1                                       # K1 once for 2 x 3 batches
    assert len(outs) == 2 * 3
    with torch.no_grad():
        want = ref(ds.get(0).x, ds.edge_index)
        l1 = sum(float(O.loss_func(ref(ds.get(t).x, ds.edge_index), ds.get(t).x, ds.get(t).target_mask))
                 for t in range(2)) * 3 / 2
    for o in outs[:3]:
        assert rel_err(o, want) <= REL_TOL
    assert abs(loss - l1) <= 1e-4 * max(1.0, abs(l1))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    a = D.train_epoch(model, ds, 50, "cuda:0", opt)
    b = D.train_epoch(model, ds, 50, "cuda:0", opt)
    assert b < a
