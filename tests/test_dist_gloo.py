"""The N > 1 path on CPU: two processes over gloo exercise member sharding and the single final
all-gather (gwen_amd/ensemble.py) with the same call sequence bench.py uses on RCCL.

The per-member "model step" is the CPU oracle here (a GPU is not available to these tests); what is
under test is the partitioning and the collective, not the kernels."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, members, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gwen_amd import ensemble
        from gwen_amd.mesh import geodesic_mesh
        from oracle import gcn_oracle as O
        torch.manual_seed(23)                                  # same weights on every rank
        mesh = geodesic_mesh(3)
        n = mesh.num_nodes
        ei = torch.from_numpy(mesh.edge_index)
        model = O.OracleGNNModel(O.OracleGNNConfig(n, n, 4, 4, 16))
        lo, hi = ensemble.member_range(members, rank, world)
        x = torch.stack([torch.randn(n, 4, generator=torch.Generator().manual_seed(23 + m))
                         for m in range(lo, hi)]) if hi > lo else torch.zeros(0, n, 4)

        def step(state):
            with torch.no_grad():
                return torch.stack([model(s, ei) for s in state]) if state.size(0) else state

        got = ensemble.ensemble_rollout(step, x, 2, members)
        # every rank must hold every member, in member order, equal to a single-process run
        full = torch.stack([torch.randn(n, 4, generator=torch.Generator().manual_seed(23 + m))
                            for m in range(members)])
        want = step(step(full))
        ok = got.shape == want.shape and torch.equal(got, want)
        # elapsed-time reduction used by bench.py: MAX over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t) == float(world)
        q.put((rank, bool(ok), tuple(got.shape)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("members", [2, 3, 4])
def test_two_ranks_gather_every_member_in_order(members):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, members, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2][0] == members for r in res)


def _forecast_worker(rank, world, port, members, q):
    """ensemble_forecast (the c5 driver: local members advanced together for n_steps, ONE gather) on two gloo
    ranks.  The step itself is the CPU oracle behind the two methods the driver needs of a model."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from gwen_amd import ensemble
        from gwen_amd.forecaster import ensemble_forecast
        from gwen_amd.mesh import geodesic_mesh
        from oracle import gcn_oracle as O

        mesh = geodesic_mesh(2)
        n = mesh.num_nodes
        ei = torch.from_numpy(mesh.edge_index)
        torch.manual_seed(23)
        w = torch.randn(3, 3) * 0.3

        class Graphs:                                   # what ensemble_forecast asks of the graphs
            def batched(self, m):
                return ("batched", m)

        class OracleStepModel:                          # ... and of the model
            def _static(self, graphs):
                return graphs

            def _step(self, x, graphs, static):
                assert graphs == static and graphs[0] == "batched"
                mloc = graphs[1]
                xs = x.view(mloc, n, 3)
                return torch.stack([torch.tanh(O.gcn_conv(xs[k], ei, w, None)) for k in range(mloc)]).view_as(x)

        lo, hi = ensemble.member_range(members, rank, world)
        xm = torch.stack([torch.randn(n, 3, generator=torch.Generator().manual_seed(7 + m)) for m in range(lo, hi)]) \
            if hi > lo else torch.zeros(0, n, 3)
        got = ensemble_forecast(OracleStepModel(), Graphs(), xm, 3, members, graphed=False)
        full = torch.stack([torch.randn(n, 3, generator=torch.Generator().manual_seed(7 + m)) for m in range(members)])
        want = full
        for _ in range(3):
            want = torch.stack([torch.tanh(O.gcn_conv(want[k], ei, w, None)) for k in range(members)])
        q.put((rank, bool(got.shape == want.shape and torch.equal(got, want)), tuple(got.shape)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("members", [2, 5])
def test_two_ranks_ensemble_forecast_gathers_once(members):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_forecast_worker, args=(r, world, port, members, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2][0] == members for r in res)
