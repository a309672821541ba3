"""Caller-side pieces of the hot path: the input producer and the loops around
``GNNModel.forward``, restated without xarray / torch-geometric / MLflow.

* ``MemberGraphDataset`` -- what ``GraphDataset`` produces (/root/reference/src/gwen/utils.py:164-211):
  nodes are ensemble members, the graph is the complete graph K_N (``erdos_renyi_graph(N, 1)``, :176),
  node features are the flattened ``height x ncells`` field of one time index (:188, :195-202), and a
  boolean ``target_mask`` marks the members after the split (:182-185, :204-206).
* ``full_graph_batches`` -- what ``NeighborLoader(data, num_neighbors=[-1, -1], batch_size=b,
  shuffle=False)`` yields on K_N (/root/reference/src/gwen/models_gnn.py:351-356): ceil(N/b) batches,
  each the WHOLE graph relabelled with the batch's seed nodes first (every node is a 1-hop neighbour of
  every seed).  Because K_N is invariant under relabelling, the relabelled ``edge_index`` is the same
  tensor object for every batch -- so the prepared graph (K1) is built once per dataset, not once per
  batch x layer as in the reference.
* ``eval_loop`` / ``train_epoch`` -- the bodies of ``eval_gnn_with_configs`` (:428-465) and
  ``train_with_configs`` (:347-376) minus process-group set-up, MLflow and the debugging switch
  ``CUDA_LAUNCH_BLOCKING`` (:320).  ``eval_loop`` returns the FULL output of every batch (the
  reference keeps only ``output[1]``, :449 -- a caller bug).
* ``extract_state_dict`` -- checkpoint shim (f4): turn any module/dict that carries the 20 reference keys
  into a plain ``state_dict`` for ``gwen_amd.GNNModel``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .mesh import complete_graph
from .models_gnn import GNNModel, loss_func


@dataclass
class GraphSample:
    x: Tensor              # [N, channels] fp32
    edge_index: Tensor     # [2, N(N-1)] int64 (shared by every sample of a dataset)
    target_mask: Tensor    # [N] bool


class MemberGraphDataset:
    """``data``: array-like ``[time, member, height, ncells]`` (what ``load_data`` returns as xarray,
    utils.py:478-520, here a numpy array or tensor).  ``split`` = number of input members
    (``member_split`` of config.json:12); the remaining members are targets."""

    def __init__(self, data, split: int, seed: Optional[int] = None):
        arr = torch.as_tensor(np.asarray(data) if not isinstance(data, Tensor) else data)
        if arr.dim() != 4:
            raise ValueError("data must be [time, member, height, ncells]")
        self.data = arr
        self.split = int(split)
        self.nodes = int(arr.size(1))
        self.edge_index = torch.from_numpy(complete_graph(self.nodes))      # utils.py:176
        rng = np.random.default_rng(seed)
        member_indices = np.arange(self.nodes)
        rng.shuffle(member_indices)                                          # utils.py:180
        self.input_indices = member_indices[: self.split]
        self.target_indices = member_indices[self.split:]
        self.channels = int(arr.size(2) * arr.size(3))                       # utils.py:188

    def __len__(self) -> int:
        return int(self.data.size(0))

    len = __len__

    def get(self, idx: int) -> GraphSample:
        x = self.data[idx].reshape(self.nodes, self.channels).to(torch.float32)
        mask = torch.zeros(self.nodes, dtype=torch.bool)
        mask[torch.from_numpy(self.target_indices)] = True
        return GraphSample(x=x, edge_index=self.edge_index, target_mask=mask)

    def __getitem__(self, idx: int) -> GraphSample:
        return self.get(idx)

    def __iter__(self) -> Iterator[GraphSample]:
        return (self.get(i) for i in range(len(self)))


def batch_permutation(num_nodes: int, start: int, batch_size: int) -> np.ndarray:
    """Node order of one loader batch: the seeds ``start .. start+batch_size-1`` first, every other
    node after them in ascending order."""
    seeds = np.arange(start, min(start + batch_size, num_nodes))
    rest = np.concatenate([np.arange(0, start), np.arange(seeds[-1] + 1, num_nodes)])
    return np.concatenate([seeds, rest])


def full_graph_batches(sample: GraphSample, batch_size: int) -> Iterator[Tuple[GraphSample, np.ndarray]]:
    """Yield ``(batch, perm)``: ``batch.x = sample.x[perm]`` etc.; ``batch.edge_index`` is the dataset's
    own K_N tensor (K_N relabelled is K_N), so every batch re-uses one prepared graph."""
    n = sample.x.size(0)
    if batch_size < 1:
        raise ValueError("batch_size must be >= 1")
    for start in range(0, n, batch_size):
        perm = batch_permutation(n, start, batch_size)
        p = torch.from_numpy(perm)
        yield GraphSample(sample.x[p], sample.edge_index, sample.target_mask[p]), perm


def _device_batches(num_nodes: int, batch_size: int, device):
    """All batch permutations of a time index and their inverses as device tensors ``[nb, N]`` -- they
    depend on (N, batch_size) only, so they are built once per loop, not once per batch."""
    if batch_size < 1:
        raise ValueError("batch_size must be >= 1")
    perms = np.stack([batch_permutation(num_nodes, s, batch_size) for s in range(0, num_nodes, batch_size)])
    inv = np.empty_like(perms)
    rows = np.arange(perms.shape[0])[:, None]
    inv[rows, perms] = np.arange(num_nodes)[None, :]
    return torch.from_numpy(perms).to(device), torch.from_numpy(inv).to(device)


def eval_loop(model: GNNModel, dataset: MemberGraphDataset, batch_size: int, device,
              loss_fn=loss_func) -> Tuple[float, List[Tensor]]:
    """Evaluation body of models_gnn.py:428-465: returns (mean loss over time indices, outputs), one
    output ``[N, C_out]`` per batch in ORIGINAL node order.

    The batches are those of ``full_graph_batches`` (seed-first permutations of the full graph), but a
    time index crosses PCIe once: the sample is moved to the device as a whole and permuted there, the
    loss is accumulated on the device and read back once -- the per-batch ``.to(device)`` and
    ``loss.item()`` of the reference's loop (models_gnn.py:358-360,:447) cost 100x the forward at its
    own shape (14 ms against 0.13 ms per batch at 125 members x 16 384 channels)."""
    model = model.to(device).eval()
    ei = dataset.edge_index.to(device)                  # one tensor object => one K1 for the whole run
    perms, invs = _device_batches(dataset.nodes, batch_size, device)
    outs: List[Tensor] = []
    running = torch.zeros((), dtype=torch.float32, device=device)
    with torch.no_grad():
        for sample in dataset:
            x_all, mask_all = sample.x.to(device), sample.target_mask.to(device)
            for b in range(perms.size(0)):
                x = x_all.index_select(0, perms[b])
                out = model(x, ei)
                running += loss_fn(out, x, mask_all.index_select(0, perms[b]))
                outs.append(out.index_select(0, invs[b]))
    return float(running) / max(len(dataset), 1), outs


def train_epoch(model: GNNModel, dataset: MemberGraphDataset, batch_size: int, device, optimizer,
                scheduler=None, loss_fn=loss_func) -> float:
    """One epoch of models_gnn.py:349-376 (zero_grad, forward, L1 on target rows, backward, step);
    data movement as in ``eval_loop`` (one copy per time index, loss read back once)."""
    model = model.to(device).train()
    ei = dataset.edge_index.to(device)
    perms, _ = _device_batches(dataset.nodes, batch_size, device)
    running = torch.zeros((), dtype=torch.float32, device=device)
    for sample in dataset:
        x_all, mask_all = sample.x.to(device), sample.target_mask.to(device)
        for b in range(perms.size(0)):
            x = x_all.index_select(0, perms[b])
            optimizer.zero_grad()
            loss = loss_fn(model(x, ei), x, mask_all.index_select(0, perms[b]))
            loss.backward()
            optimizer.step()
            if scheduler is not None:
                scheduler.step()
            running += loss.detach()
    return float(running) / max(len(dataset), 1)


_LAYERS = [f"down_conv_layers.conv{i}" for i in range(1, 6)] + [f"up_conv_layers.upconv{i}" for i in range(1, 6)]
REFERENCE_KEYS = [f"conv_layers.{l}.{p}" for l in _LAYERS for p in ("bias", "lin.weight")]


def extract_state_dict(source) -> Dict[str, Tensor]:
    """Checkpoint shim: accept a reference ``GNNModel`` (un-pickled in an environment that has
    torch-geometric), any ``nn.Module`` with the same parameter names, a plain ``state_dict``, or a
    DistributedDataParallel-style dict with a ``module.`` prefix, and return exactly the 20 tensors
    ``gwen_amd.GNNModel.load_state_dict(..., strict=True)`` expects."""
    sd = source.state_dict() if hasattr(source, "state_dict") else dict(source)
    out: Dict[str, Tensor] = {}
    for key in REFERENCE_KEYS:
        for cand in (key, "module." + key, "model." + key):
            if cand in sd:
                out[key] = sd[cand].detach().to(torch.float32).cpu().clone()
                break
        else:
            raise KeyError(f"checkpoint lacks {key!r}")
    return out


def config_from_state_dict(sd: Dict[str, Tensor]):
    """Recover ``GNNConfig`` widths (channels_in, channels_out, hidden_feats) from the tensor shapes."""
    from .models_gnn import GNNConfig
    w1 = sd["conv_layers.down_conv_layers.conv1.lin.weight"]
    w5 = sd["conv_layers.up_conv_layers.upconv5.lin.weight"]
    return GNNConfig(nodes_in=0, nodes_out=0, channels_in=int(w1.size(1)), channels_out=int(w5.size(0)),
                     hidden_feats=int(w1.size(0)))
