#!/usr/bin/env python3
"""bench.py -- GWEN GCNConv-stack hot path on MI355X: mesh edges/s (message+aggregate).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "c2"): synthetic ICON-CH1-scale geodesic mesh, nu = 100 ->
N = 100 002 nodes, E = 600 000 directed edges; GNNModel(channels 64 -> hidden 64 -> 64), fp32,
random-init weights (seed 23), 1 ensemble member per GPU.  One STEP = one GNNModel.forward of every
local member = 6 GCNConv layers = 6 message+aggregate passes over the E mesh edges (plus their dense
projections, bias and ReLU).  value = members * 6 * E * K / t  [edge passes per second, whole job].
Inputs, weights and the prepared graph are resident in HBM before the timed region.  With N > 1
members are sharded one per rank (weak scaling), there is no collective on the data path, and the
single RCCL all-gather of the final states sits INSIDE the timed region, after the last step.

Two extra objects ride on the JSON line (see DESIGN.md "Measurement"):
  roofline     -- the dominant kernel (by summed time) of the timed region: algorithmic bytes per
                  launch / its average duration from hipEvents the launcher records around every
                  kernel launch (on the launch stream) inside the timed region, against 8 TB/s.
  cpu_baseline -- the torch oracle (kind "port": the reference's PyG is not installable) timed on
                  this host's cores for the same workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--nu", type=int, default=100, help="mesh frequency: N = 10 nu^2 + 2")
    p.add_argument("--channels", type=int, default=64)
    p.add_argument("--hidden", type=int, default=64)
    p.add_argument("--members-per-gpu", type=int, default=1)
    p.add_argument("--reorder", default="morton", choices=["none", "morton"])
    p.add_argument("--order", default="auto", choices=["auto", "unfused"],
                   help="auto: K4 fused layer where the widths allow; unfused: K3 + K2 per layer")
    p.add_argument("--event-stride", type=int, default=40,
                   help="record per-kernel hipEvents on every n-th timed step (each pair of records "
                        "opens a ~10 us gap on the stream, so instrumenting every step would slow the "
                        "steps being timed)")
    p.add_argument("--graph", action="store_true",
                   help="replay a captured hipGraph per step instead of issuing the 6 launches from the "
                        "C launcher (measured slower here: one graph launch costs more than 6 direct ones)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-edge-mlp", action="store_true",
                   help="skip the side measurement of the InteractionNet edge-MLP kernel (K6)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    return p.parse_args()


def algorithmic_bytes(kind, n, e, fin, fout):
    """B_alg = gathered source rows + self-loop row + output row + int32 col + fp32 weight
    per stored entry + rowptr + (K3: x read, W read, h written)."""
    if kind == "propagate":      # K2 at width F = fin = fout
        f = fin
        return 4 * f * (e + 2 * n) + 8 * e + 8 * n
    if kind == "linear":         # K3: x read, h written, W read
        return 4 * n * (fin + fout) + 4 * fin * fout
    if kind == "chain":          # K5: gather at fin (E edges + self-loop), store at fout, indices
        return 4 * fin * (e + n) + 4 * fout * n + 8 * e + 8 * n
    if kind == "layer":          # K4: gather at fin (E edges + self-loop), store at fout, indices, W
        return 4 * fin * (e + n) + 4 * fout * n + 8 * e + 8 * n + 4 * fin * fout
    raise KeyError(kind)


def edge_mlp_side_measurement(mesh, f, dev, launches=30):
    """K6 (gwen_mlp2_f32: gathers + edge MLP + residual + in-order sum to targets) on the mesh's edges
    at width f: torch events on the launch stream around `launches` back-to-back launches."""
    from gwen_amd import ops
    from gwen_amd.interaction import InteractionNet, interaction_graph, mlp2
    g = interaction_graph(torch.from_numpy(mesh.edge_index).to(dev), mesh.num_nodes, mesh.num_nodes)
    n, e = mesh.num_nodes, g.num_edges
    torch.manual_seed(23)
    net = InteractionNet(f).to(dev)
    x = torch.randn(n, f, device=dev)
    ef = torch.randn(e, f, device=dev)
    with torch.no_grad():
        we, wa, wn, bn = net._weight_blocks()
        p = ops.linear(x, wn, bn, exact=False)
        run = lambda: mlp2(ef, we, net.edge_mlp[2].weight, net.edge_mlp[2].bias, g1=p[:, :f], idx1=g.src,   # noqa: E731
                           g2=p[:, f:2 * f], idx2=g.dst, res=ef, graph=g)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            run()
        b.record()
        torch.cuda.synchronize()
    t = a.elapsed_time(b) / launches * 1e-3
    b_alg = 4 * f * (4 * e + n) + 8 * e + 4 * n      # e read, e' written, two gathered rows, agg, indices
    traffic = None                                    # HBM-side bytes per launch from the committed PMC passes
    try:
        tf = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"]
        if (n, e) == (100002, 600000):
            traffic = tf[f"k_mlp2<{f}, 2, 2, true>"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        traffic = None
    return {"workload": f"InteractionNet edge kernel (K6), F={f}, same mesh: gathers + 2-layer edge MLP + "
                        f"residual + in-order sum to targets, one launch", "edges": e,
            "us_per_launch": round(t * 1e6, 1), "edge_updates_per_s": round(e / t),
            "roofline": {"bound": "hbm" if f <= 64 else "mfma", "algorithmic_bytes": b_alg,
                         "achieved_GBs": round(b_alg / t / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                         "frac_hbm": round(b_alg / t / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bf16_tflops_issued": round(12 * f * f * e / t / 1e12, 1)}}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import gwen_amd
    from gwen_amd import ensemble

    # ---- synthetic inputs, resident in HBM ------------------------------------------------------
    mesh = gwen_amd.geodesic_mesh(args.nu, reorder=None if args.reorder == "none" else args.reorder)
    n, e = mesh.num_nodes, mesh.num_edges
    c, h = args.channels, args.hidden
    m_local = args.members_per_gpu
    members = m_local * world
    lo, hi = ensemble.member_range(members, rank, world)
    torch.manual_seed(23)                                            # config.json:14
    model = gwen_amd.GNNModel(gwen_amd.GNNConfig(n, n, c, c, h))
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.dim() == 1:
                p_.normal_(0.0, 0.1)                                 # exercise the bias path
    model = model.to(dev).eval()
    if args.order == "unfused":
        for mod in model.modules():
            if isinstance(mod, gwen_amd.GCNConv):
                mod.order = "aggregate_first" if mod.in_channels < mod.out_channels else "transform_first"
    x = torch.stack([torch.randn(n, c, generator=torch.Generator().manual_seed(23 + m))
                     for m in range(lo, hi)]).to(dev)
    if m_local == 1:
        x = x[0]
    edge_index = torch.from_numpy(mesh.edge_index).to(dev)
    graph = model.prepare(edge_index, n)                              # K1, once, outside the timed region
    layers = 6
    widths = [(c, h), (h, h // 2), (h // 2, h // 4), (h // 4, h // 2), (h // 2, h), (h, c)]

    # ---- the whole stack behind one C call; hipEvents around every kernel launch ----------------
    plan = gwen_amd.StackForward(model.stack(), graph)
    out = plan.run(x)                                                 # allocates scratch + output
    stride = max(1, args.event_stride)                                # events on every stride-th step
    n_sets = (args.steps + stride - 1) // stride
    ev_sets = [gwen_amd.KernelEvents(2 * layers) for _ in range(n_sets)]

    graphed = gwen_amd.GraphedForward(plan, x) if args.graph else None
    if graphed is not None:
        out = graphed.out

    def step(ev=None):
        # every step is the same work; steps that carry hipEvents are issued launch by launch (events
        # cannot be read back from inside a graph), all others replay the captured hipGraph
        if ev is not None or graphed is None:
            return plan.run(x, out=out, events=ev)
        return graphed()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    if world > 1:   # RCCL sets its rings/channels up on the first collective: keep that out of the timing
        ensemble.gather_members(out if out.dim() == 3 else out.unsqueeze(0), members)
    torch.cuda.synchronize()
    barrier()

    # ---- timed region ---------------------------------------------------------------------------
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev_sets[i // stride] if i % stride == 0 else None)
    final = out if out.dim() == 3 else out.unsqueeze(0)
    gathered = ensemble.gather_members(final, members) if world > 1 else final
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    assert gathered.shape[0] == members and torch.isfinite(gathered).all()

    # a bracket = kernel + one event record's worth of stream time: calibrate the latter on empty
    # brackets and take it off (agrees with rocprofv3's kernel-only durations within 0.4 us)
    ev_over = gwen_amd.event_bracket_overhead(dev)
    summ = {}
    for evs in ev_sets:
        for kind, layer, fin, fout, sec in evs.durations():
            cnt, tot = summ.get((kind, fin, fout), (0, 0.0))
            summ[(kind, fin, fout)] = (cnt + 1, tot + max(sec - ev_over, 0.0))

    # ---- roofline of the dominant kernel (events recorded inside the timed region) ---------------
    dom_key = max(summ, key=lambda k: summ[k][1])
    launches, total_s = summ[dom_key]
    kind, fin, fout = dom_key
    b_alg = algorithmic_bytes(kind, n, e, fin, fout) * m_local
    avg_s = total_s / launches
    achieved = b_alg / avg_s / 1e9
    # HBM-side bytes per launch of that kernel from the committed PMC passes (tools/profile_bench.sh +
    # tools/collect_traffic.py; FETCH_SIZE and WRITE_SIZE need separate rocprofv3 passes)
    traffic = None
    try:
        tf = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"]
        tag = {"layer": f"k_layer<{fin}, {fout},", "chain": f"k_chain<{fin},", "linear": "k_linear<",
               "propagate": "k_propagate<"}[kind]
        hits = [v["hbm_bytes_per_launch"] for k_, v in tf.items()
                if k_.startswith(tag) and (kind != "chain" or k_.split(",")[1].strip() != "0"
                                           and fout in (int(k_.split(",")[1]), int(k_.split(",")[2])))]
        if hits and (n, e, c, h, m_local) == (100002, 600000, 64, 64, 1):
            traffic = hits[0]
    except (OSError, KeyError, ValueError):
        traffic = None
    roofline = {
        "bound": "hbm", "kernel": f"{kind}_f32[{fin}->{fout}]", "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic, "algorithmic_bytes_per_launch": b_alg, "avg_launch_us": round(avg_s * 1e6, 2),
        "launches": launches, "event_record_overhead_us": round(ev_over * 1e6, 2),
        "all_kernels_us": {f"{k[0]}[{k[1]}->{k[2]}]": round(v[1] / v[0] * 1e6, 2) for k, v in sorted(summ.items())},
    }

    value = members * layers * e * args.steps / elapsed
    line = {
        "metric": "mesh edges/s (message+aggregate)", "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c2: geodesic mesh nu={args.nu} N={n} E={e}, GNNModel forward "
                               f"C={c} H={h} (6 GCNConv layers), {m_local} member/GPU",
                   "nodes": n, "edges": e, "channels": c, "hidden": h, "layers": layers,
                   "members": members, "node_order": args.reorder, "kernel_order": args.order, "hip_graph": bool(args.graph),
                   "parallelism": f"ensemble members sharded 1 rank = {m_local} member(s); one all-gather at end"},
        "members_per_s": members * args.steps / elapsed,
        "roofline": roofline,
    }

    # ---- side measurement (outside the timed region, N = 1 only): the InteractionNet edge-MLP kernel
    # K6 on the same mesh at the same width -- the block BASELINE.json's north_star names; the headline
    # value above stays the reference's own layer (GCNConv) -------------------------------------------
    if rank == 0 and world == 1 and not args.no_edge_mlp and h in (32, 64, 128, 256):
        line["edge_mlp_block"] = edge_mlp_side_measurement(mesh, h, dev)

    # ---- CPU baseline: the torch oracle on this host's cores (rank 0, N = 1 only) ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import gcn_oracle as O
        ref = O.OracleGNNModel(O.OracleGNNConfig(n, n, c, c, h))
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=True)
        xc = (x if x.dim() == 2 else x[0]).cpu()
        eic = torch.from_numpy(mesh.edge_index)
        cores = torch.get_num_threads()
        with torch.no_grad():
            tw = time.perf_counter(); yc = ref(xc, eic); one = time.perf_counter() - tw   # warm-up
            reps = max(1, min(20, int(args.cpu_seconds / max(one, 1e-3))))
            ts = []
            for _ in range(reps):
                tw = time.perf_counter(); yc = ref(xc, eic); ts.append(time.perf_counter() - tw)
        med = sorted(ts)[len(ts) // 2]
        got = (out if out.dim() == 2 else out[0]).cpu()
        err = float((got - yc).abs().max() / yc.abs().max())
        cpu_model = "unknown CPU"
        try:
            with open("/proc/cpuinfo") as fh:
                cpu_model = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
        except (OSError, StopIteration):
            pass
        line["cpu_baseline"] = {
            "value": layers * e / med, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{reps} full GNNModel.forward passes of the same c2 workload (1 member), median "
                      f"{med*1e3:.1f} ms, torch {torch.__version__} CPU ({cores} threads on {cpu_model}), "
                      f"oracle/gcn_oracle.py",
            "gpu_vs_oracle_rel_err": err,
        }
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
