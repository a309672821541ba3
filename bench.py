#!/usr/bin/env python3
"""bench.py -- GWEN GCNConv-stack hot path on MI355X: mesh edges/s (message+aggregate).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, one per GPU: the reference's
                                                            mp.spawn, /root/reference/src/gwen/train_gnn.py:144-152)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --workload c5 [--gpus N]               (BASELINE configs[4]: 4 members per GPU, 4-step
                                                            grid->mesh->grid rollout, members/s incl. the one gather)

Headline workload (BASELINE.json configs[1], "c2"): synthetic ICON-CH1-scale geodesic mesh, nu = 100 ->
N = 100 002 nodes, E = 600 000 directed edges; GNNModel(channels 64 -> hidden 64 -> 64), fp32 storage,
random-init weights (seed 23), 1 ensemble member per GPU.  One STEP = one GNNModel.forward of every
local member = 6 GCNConv layers = 6 message+aggregate passes over the E mesh edges (plus their dense
projections, bias and ReLU).  value = members * 6 * E * K / t  [edge passes per second, whole job].
Inputs, weights and the prepared graph are resident in HBM before the timed region.  With N > 1
members are sharded one per rank (weak scaling), there is no collective on the data path, and the
single RCCL all-gather of the final states sits INSIDE the timed region, after the last step.

Extra objects on the JSON line (DESIGN.md "Measurement"):
  roofline      -- dominant kernel of the c2 step.  c2's working set (3 x 25.6 MB) lives in the L2s and the
                   256 MiB Infinity Cache, so the ceiling that binds is the L2 ("bound": "l2", peak 34.5
                   TB/s, MI355X_MICROARCH.md): achieved = bytes the kernel pulls through L1/L2 (SURVEY 8(d)'s
                   algorithmic bytes) / its mean duration from >= 10 hipEvent samples.  The compulsory
                   (HBM-side) figure rides along as frac_hbm_compulsory.
  hbm_leg       -- the same path where HBM IS the bound (N = 1 only, outside the timed region): BASELINE
                   config c3's stack (4 chained 256 -> 256 GCN layers + ReLU) on 4 members per GPU -- the
                   per-GPU load of config c5 -- 410 MB per activation buffer, far beyond the Infinity Cache.
                   Its roofline: compulsory bytes (every input, output, index and weight byte once) / mean
                   kernel duration / 8 TB/s; traffic = PMC bytes from the committed rocprofv3 passes.
  hbm_leg_64ch  -- the same stack at c2's 64 channels on 16 members (819 MB of activations per layer): the layer
                   kernel where the 3xbf16 matrix work is a quarter of the 256-channel one per byte.
  edge_mlp_block, edge_mlp_block_8_members -- the InteractionNet edge-MLP + aggregation kernel (K6, the block
                   BASELINE's north_star names) on the same mesh at the same width, one member and 8 members
                   batched: us per launch and the compulsory-bytes HBM fraction.
  exact_f32     -- the c2 step with every contraction on the fp32-input MFMA (order "fused_exact").
  cpu_baseline  -- the torch oracle (kind "port": the reference's PyG is not installable) timed on this
                   host's cores, all cores and one thread, rank 0, N = 1 only.
  allgather     -- N > 1: duration and bus bandwidth of the one collective, ranks seen.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0          # same guide, "L2 (per XCD)": ~34.5 TB/s aggregate
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: "~2.5 PF dense" bf16 MFMA
L2_GATHER_GBS = 17800.0        # same guide, "Indexed rows": rows served by the XCD's L2, 16.8-18.8 TB/s chip-wide
MIN_SAMPLES = 10


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="default: 200 (c2), 10 (c5)")
    p.add_argument("--warmup", type=int, default=None, help="default: 20 (c2), 2 (c5)")
    p.add_argument("--workload", default="c2", choices=["c2", "c5"],
                   help="c2: BASELINE configs[1], GNNModel forward on the nu=100 mesh, edges/s (the headline); "
                        "c5: BASELINE configs[4], InteractionNet grid->mesh->grid forecaster, 4 members per GPU, "
                        "4-step autoregressive rollout + ONE all-gather, members/s")
    p.add_argument("--launch-dry-run", action="store_true",
                   help="with --gpus N > 1 and no WORLD_SIZE in the environment: print the N child command lines "
                        "and their rank environments as one JSON object and exit without starting them")
    p.add_argument("--rehearse-gloo", action="store_true",
                   help="REHEARSAL of the N > 1 code path on a box with fewer GPUs than ranks: every rank uses GPU 0 and "
                        "the process group is gloo (RCCL refuses two ranks on one device), the gather goes through the "
                        "host.  The line says so (allgather.backend = gloo, config.rehearsal); not a measurement")
    p.add_argument("--rollout-steps", type=int, default=4, help="c5: autoregressive steps per rollout")
    p.add_argument("--c5-channels", type=int, default=256, help="c5: grid channels = hidden width")
    p.add_argument("--c5-blocks", type=int, default=4, help="c5: mesh->mesh processor blocks")
    p.add_argument("--c5-members-per-gpu", type=int, default=4)
    p.add_argument("--prewarm-ms", type=float, default=100.0,
                   help="untimed steady-state run-in before the W warm-up steps (the first ms after an idle gap "
                        "run at ramping clocks)")
    p.add_argument("--nu", type=int, default=100, help="mesh frequency: N = 10 nu^2 + 2")
    p.add_argument("--channels", type=int, default=64)
    p.add_argument("--hidden", type=int, default=64)
    p.add_argument("--members-per-gpu", type=int, default=1)
    p.add_argument("--reorder", default="hilbert", choices=["none", "morton", "hilbert"])
    p.add_argument("--order", default="auto", choices=["auto", "bf16x3", "unfused", "fused_exact"],
                   help="auto: fused kernels, bf16x6 contraction (fp32-class, the library default); bf16x3: the same "
                        "kernels on the faster two-image split; fused_exact: fp32-input MFMA; unfused: K3 + K2 per layer")
    p.add_argument("--event-stride", type=int, default=40,
                   help="record per-kernel hipEvents on every n-th timed step (each pair of records "
                        "opens a ~10 us gap on the stream, so instrumenting every step would slow the "
                        "steps being timed); more instrumented steps follow the timed region until "
                        f"every kernel has {MIN_SAMPLES} samples")
    p.add_argument("--graph", action="store_true",
                   help="replay a captured hipGraph per step instead of issuing the launches from the C "
                        "launcher (measured slower here: one graph launch costs more than 6 direct ones)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-live-traffic", action="store_true",
                   help="do not run the two rocprofv3 --pmc child passes at the start (roofline.traffic then comes from "
                        "the committed passes, profiles/traffic.json)")
    p.add_argument("--live-traffic-limit", type=float, default=150.0, help="time limit of each PMC child pass, seconds")
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--allow-variant", action="store_true",
                   help="run although GWEN_HIP_LIB points at an experimental build of the library (the line then "
                        "carries library.variant = true; such a line is not a measurement of the product)")
    p.add_argument("--rank-timeout", type=float, default=1500.0,
                   help="self-launch: wall-clock limit of the ranks; past it they are terminated, then killed")
    p.add_argument("--no-edge-mlp", action="store_true",
                   help="skip the side measurement of the InteractionNet edge-MLP kernel (K6)")
    p.add_argument("--edge-mlp-members", type=int, default=8,
                   help="members of the batched edge-MLP side measurement (0: skip -- the PMC passes do, so that a "
                        "kernel name averages over one launch shape)")
    p.add_argument("--edge-mlp-skip-single", action="store_true",
                   help="only the batched edge-MLP side measurement (the PMC pass of the 8-member launch shape)")
    p.add_argument("--no-hbm-leg", action="store_true")
    p.add_argument("--no-exact", action="store_true")
    p.add_argument("--hbm-members", type=int, default=4)
    p.add_argument("--hbm-channels", type=int, default=256)
    p.add_argument("--hbm-members-narrow", type=int, default=16, help="members of the 64-channel HBM leg")
    p.add_argument("--hbm-layers", type=int, default=4)
    p.add_argument("--hbm-steps", type=int, default=12)
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of each CPU baseline leg")
    a = p.parse_args()
    if a.steps is None:
        a.steps = 200 if a.workload == "c2" else 10
    if a.warmup is None:
        a.warmup = 20 if a.workload == "c2" else 2
    return a


# ---- self-launch: `python bench.py --gpus N` starts its own ranks ------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_plan(n_ranks, argv, port=None, base_env=None):
    """The N child processes `python bench.py --gpus N` starts when no launcher set WORLD_SIZE: one per GPU, each
    a FRESH interpreter running this file with the parent's flags and the rank environment torch.distributed.run
    would give it (the reference starts its ranks itself too: /root/reference/src/gwen/train_gnn.py:144-152,
    rendezvous by MASTER_ADDR / MASTER_PORT, models_gnn.py:321-322).  Returns [(argv, env-additions)]."""
    port = port or _free_port()
    child_argv = [a for a in argv if a != "--launch-dry-run"]
    plan = []
    for r in range(n_ranks):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
               "HSA_ENABLE_IPC_MODE_LEGACY": (base_env or os.environ).get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
        plan.append(([sys.executable, os.path.abspath(__file__)] + child_argv, env))
    return plan


def run_plan(plan, poll_s=0.2, timeout_s=1500.0, grace_s=10.0):
    """Start every child, wait for all; the first non-zero exit code ends the others (exact PIDs) and is returned.
    A rank stuck in a collective can not hang the parent: past `timeout_s` of wall clock -- or `grace_s` after the others
    were told to terminate -- the remaining children are terminated, then killed, and the result is non-zero (124).
    The parent never touches the GPU, so nothing that has initialised HIP is ever re-executed."""
    procs = []
    for argv, env in plan:
        procs.append(subprocess.Popen(argv, env={**os.environ, **env}))
    rc = 0
    live = list(procs)
    t0 = time.monotonic()
    term_at = None                    # when the live children were sent SIGTERM
    while live:
        time.sleep(poll_s)
        for p_ in list(live):
            code = p_.poll()
            if code is None:
                continue
            live.remove(p_)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                for q_ in live:
                    q_.terminate()
                term_at = time.monotonic()
        now = time.monotonic()
        if live and term_at is None and now - t0 > timeout_s:
            rc = rc or 124
            for q_ in live:
                q_.terminate()
            term_at = now
        elif live and term_at is not None and now - term_at > grace_s:
            for q_ in live:
                q_.kill()                                   # exact PIDs of our own children
            term_at = now
    return rc


def algorithmic_bytes(kind, n, e, fin, fout):
    """bytes the kernel pulls through the L1/L2 path = gathered source rows + self-loop row +
    output row + int32 col + fp32 weight per stored entry + rowptr (+ K3: x read, W read, h written)."""
    if kind == "propagate":      # K2 at width F = fin = fout
        f = fin
        return 4 * f * (e + 2 * n) + 8 * e + 8 * n
    if kind == "linear":         # K3: x read, h written, W read
        return 4 * n * (fin + fout) + 4 * fin * fout
    if kind == "chain":          # K5: gather at fin (E edges + self-loop), store at fout, indices
        return 4 * fin * (e + n) + 4 * fout * n + 8 * e + 8 * n
    if kind in ("layer", "wide"):    # K4 / K8: gather at fin, store at fout, indices, W
        return 4 * fin * (e + n) + 4 * fout * n + 8 * e + 8 * n + 4 * fin * fout
    raise KeyError(kind)


def compulsory_bytes(kind, n, e, fin, fout, members=1):
    """Every input, output, index and weight byte exactly once: what must cross the HBM interface when
    nothing is resident."""
    idx = 8 * (e + n) + 4 * n                        # col + val per stored entry, rowptr
    if kind == "linear":
        return 4 * members * n * (fin + fout) + 4 * fin * fout
    if kind == "propagate":
        return 8 * members * n * fin + idx
    return 4 * members * n * (fin + fout) + idx + 4 * fin * fout


PREWARM_S = 0.1


def prewarm(fn, seconds=None):
    """Keep the GPU busy with the work about to be measured for `seconds` (untimed): after an idle gap -- tensors
    made on the host, a finished leg -- the first ~5-40 ms of launches run 8-20 % slow (clock ramp; measured with
    tools/experiments/warm_ramp.py).  Every measurement below starts from the steady state."""
    seconds = PREWARM_S if seconds is None else seconds
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()


# ---- HBM traffic from the PMC counters -------------------------------------------------------------------------------
# Collected LIVE at the start of a default run: two short child processes of this file under `rocprofv3 --pmc`
# (FETCH_SIZE, then WRITE_SIZE -- they can not share a pass on gfx950 -- and never together with a trace domain), each
# launching every kernel the line quotes a few times (--pmc-child), BEFORE this process touches the GPU.  The
# corrections are the guide's (MI355X_MICROARCH.md): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream
# (x 2), both counters are in KiB.  If rocprofv3 is missing, fails or runs out of its time limit, the committed passes of
# the same command (profiles/traffic.json, tools/profile_bench.sh + tools/collect_traffic.py) are used and the line says so.
LIVE_TRAFFIC = None            # {"kernels": {name: {...}}} once collected
TRAFFIC_SOURCE = "profiles/traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                 "gfx950 corrections applied)"
KERNEL_RE = r"(k_(?:layer|chain|gather|propagate|linear_split|linear|wide|mlp2r|mlp2)<[^>]*>)"


def _pmc_pass(counter, child_argv, tmp, limit_s):
    """One `rocprofv3 --pmc <counter>` child pass; returns ({kernel: [values in dispatch order]}, None) or (None, why)."""
    import csv
    import glob
    import re
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out_dir = os.path.join(tmp, counter)
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--", sys.executable,
           os.path.abspath(__file__)] + child_argv
    env = {**os.environ, "TMPDIR": tmp}
    try:
        r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=limit_s)
    except (subprocess.TimeoutExpired, OSError) as exc:
        return None, f"{type(exc).__name__}"
    if r.returncode != 0:
        return None, f"exit code {r.returncode}"
    rows = []
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == counter:
                m = re.search(KERNEL_RE, row["Kernel_Name"])
                if m:
                    rows.append((int(row.get("Dispatch_Id", 0)), m.group(1), float(row["Counter_Value"])))
    rows.sort()
    per = {}
    for _, name, val in rows:
        per.setdefault(name, []).append(val)
    return (per, None) if per else (None, "no counter rows")


def collect_live_traffic(args):
    """Fills LIVE_TRAFFIC / TRAFFIC_SOURCE (see above).  Runs before anything in this process initialises the GPU."""
    global LIVE_TRAFFIC, TRAFFIC_SOURCE
    import tempfile
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="gwen_pmc_", dir="/tmp")
    child = ["--pmc-child", "--nu", str(args.nu), "--channels", str(args.channels), "--hidden", str(args.hidden),
             "--reorder", args.reorder, "--hbm-members", str(args.hbm_members), "--hbm-channels", str(args.hbm_channels),
             "--hbm-members-narrow", str(args.hbm_members_narrow), "--hbm-layers", str(args.hbm_layers),
             "--edge-mlp-members", str(args.edge_mlp_members)]
    fetch, why = _pmc_pass("FETCH_SIZE", child, tmp, args.live_traffic_limit)
    write, why2 = (None, why) if fetch is None else _pmc_pass("WRITE_SIZE", child, tmp, args.live_traffic_limit)
    try:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    except OSError:
        pass
    if fetch is None or write is None:
        TRAFFIC_SOURCE += f"; the live passes of this run failed ({why or why2})"
        return
    n1 = 3                          # the child launches the edge block 3 times on one member, then on the batch (below)
    kernels = {}
    for name in sorted(set(fetch) & set(write)):
        f_, w_ = fetch[name], write[name]
        groups = [("", f_, w_)]
        if name.startswith("k_mlp2") and len(f_) > n1 and len(w_) > n1 and args.edge_mlp_members > 1:
            groups = [("", f_[:n1], w_[:n1]), (f"@{args.edge_mlp_members}members", f_[n1:], w_[n1:])]
        for suffix, fv, wv in groups:
            fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
            kernels[name + suffix] = {"FETCH_SIZE_KiB_raw": round(fm, 1), "WRITE_SIZE_KiB": round(wm, 1),
                                      "launches_sampled": [len(fv), len(wv)],
                                      "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
    LIVE_TRAFFIC = {"kernels": kernels}
    TRAFFIC_SOURCE = (f"LIVE: two rocprofv3 --pmc child passes of this run (FETCH_SIZE x 2 per the guide's gfx950 note, "
                      f"WRITE_SIZE; {time.perf_counter() - t0:.0f} s, before the timed region and before this process "
                      f"touched the GPU)")


def pmc_child(args):
    """--pmc-child: launch every kernel the line quotes a few times and exit (the process rocprofv3 --pmc watches)."""
    import gwen_amd
    from gwen_amd import ops
    from gwen_amd.interaction import InteractionNet, interaction_graph, mlp2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    mesh = gwen_amd.geodesic_mesh(args.nu, reorder=None if args.reorder == "none" else args.reorder)
    n, c, h = mesh.num_nodes, args.channels, args.hidden
    torch.manual_seed(23)
    model = gwen_amd.GNNModel(gwen_amd.GNNConfig(n, n, c, c, h)).to(dev).eval()
    ei = torch.from_numpy(mesh.edge_index).to(dev)
    graph = model.prepare(ei, n)
    x = torch.randn(n, c, device=dev)
    plan = gwen_amd.StackForward(model.stack(), graph)
    for _ in range(6):
        plan.run(x)
    for f, m in ((args.hbm_channels, args.hbm_members), (c, args.hbm_members_narrow)):
        xs = torch.randn(m, n, f, device=dev)
        for order in ("auto", "auto_x3", "auto_x6"):
            layers = []
            for _ in range(args.hbm_layers):
                conv = gwen_amd.GCNConv(f, f).to(dev)
                layers.append((conv.lin.weight.detach(), conv.bias.detach(), True, order))
            st = gwen_amd.StackForward(layers, graph)
            for _ in range(2):
                st.run(xs)
        del xs
    if h in (32, 64, 128, 256):
        for members in ((1, args.edge_mlp_members) if args.edge_mlp_members > 1 else (1,)):
            g = interaction_graph(ei, n, n)
            if members > 1:
                g = g.batched(members)
            net = InteractionNet(h).to(dev)
            xe = torch.randn(g.num_dst, h, device=dev)
            ef = torch.randn(g.num_edges, h, device=dev)
            with torch.no_grad():
                we, wa, wn, bn = net._weight_blocks()
                p_ = ops.linear(xe, wn, bn, exact=False)
                for _ in range(3):                                    # (collect_live_traffic: n1 = 3)
                    mlp2(ef, we, net.edge_mlp[2].weight, net.edge_mlp[2].bias, g1=p_[:, :h], idx1=g.src,
                         g2=p_[:, h:2 * h], idx2=g.dst, res=ef, graph=g)
    torch.cuda.synchronize()


def pmc_traffic(tag_prefixes):
    """HBM-side bytes per launch of a kernel: from this run's own PMC passes (collect_live_traffic) when they
    succeeded, else from the committed ones (profiles/traffic.json)."""
    if LIVE_TRAFFIC is not None:
        tf = LIVE_TRAFFIC["kernels"]
    else:
        try:
            tf = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"]
        except (OSError, KeyError, ValueError):
            return None
    for k_, v in tf.items():
        for t in tag_prefixes:
            pre, suf = t if isinstance(t, tuple) else (t, "")
            if k_.startswith(pre) and k_.endswith(suf) and ("@" in suf or "@" not in k_):
                return v["hbm_bytes_per_launch"]
    return None


def kernel_tags(kind, fin, fout, ns):
    """Name patterns (prefix, suffix) of the kernel a launcher `kind` ran, as rocprofv3 prints it; ns = images per
    operand (2: bf16x3, 3: bf16x6, 0: fp32 MFMA, "f16": K8's two scaled fp16 images) -- template arguments of every
    contracting kernel (k_wide ends in <.., images, f16x3>)."""
    wide_suffix = ", 2, true>" if ns == "f16" else f", {ns}, false>"
    ns = 3 if ns == "f16" else ns              # every other kernel runs bf16x6 under the default precision
    return {"wide": [(f"k_wide<{fin}, {fout},", wide_suffix)], "layer": [(f"k_layer<{fin}, {fout}, {ns},", "")],
            "propagate": ["k_propagate<"], "linear": ["k_linear"],
            "chain": [(f"k_chain<{fin},", f", {ns}>"), "k_gather<"]}[kind]


class Sampler:
    """Per-kernel durations from hipEvents the launcher records around every launch."""

    def __init__(self, ga, max_launches):
        self.ga, self.max_launches = ga, max_launches
        self.sets = []

    def new(self):
        ev = self.ga.KernelEvents(self.max_launches)
        self.sets.append(ev)
        return ev

    def summary(self, dev):
        over = self.ga.event_bracket_overhead(dev)
        summ = {}
        for evs in self.sets:
            for kind, layer, fin, fout, sec in evs.durations():
                cnt, tot = summ.get((kind, fin, fout), (0, 0.0))
                summ[(kind, fin, fout)] = (cnt + 1, tot + max(sec - over, 0.0))
        return summ, over

    def min_samples(self):
        counts = {}
        for evs in self.sets:
            for kind, layer, fin, fout, _ in evs.durations():
                counts[(kind, fin, fout)] = counts.get((kind, fin, fout), 0) + 1
        return min(counts.values()) if counts else 0


def edge_mlp_side_measurement(mesh, f, dev, launches=30, members=1):
    """K6 (gwen_mlp2_f32: gathers + edge MLP + residual + in-order sum to targets) on the mesh's edges
    at width f: torch events on the launch stream around `launches` back-to-back launches.  members > 1: the
    block-diagonal graph of that many members (the c5 driver's per-GPU batch), far beyond the Infinity Cache."""
    from gwen_amd import ops
    from gwen_amd.interaction import InteractionNet, interaction_graph, mlp2
    g = interaction_graph(torch.from_numpy(mesh.edge_index).to(dev), mesh.num_nodes, mesh.num_nodes)
    if members > 1:
        g = g.batched(members)
    n, e = g.num_dst, g.num_edges
    torch.manual_seed(23)
    net = InteractionNet(f).to(dev)
    x = torch.randn(n, f, device=dev)
    ef = torch.randn(e, f, device=dev)
    with torch.no_grad():
        we, wa, wn, bn = net._weight_blocks()
        p = ops.linear(x, wn, bn, exact=False)
        run = lambda: mlp2(ef, we, net.edge_mlp[2].weight, net.edge_mlp[2].bias, g1=p[:, :f], idx1=g.src,   # noqa: E731
                           g2=p[:, f:2 * f], idx2=g.dst, res=ef, graph=g)
        prewarm(run)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            run()
        b.record()
        torch.cuda.synchronize()
    t = a.elapsed_time(b) / launches * 1e-3
    b_l2 = 4 * f * (4 * e + n) + 8 * e + 4 * n        # e read, e' written, two gathered rows, agg, indices
    b_comp = 4 * f * (2 * e + 2 * n) + 12 * e + 4 * n  # e, e' once; x, agg once; src/dst/rowptr
    return {"workload": f"InteractionNet edge kernel (K6), F={f}, same mesh x {members} member(s): gathers + 2-layer "
                        f"edge MLP + residual + in-order sum to targets, one launch", "edges": e,
            "us_per_launch": round(t * 1e6, 1), "edge_updates_per_s": round(e / t),
            "roofline": {"bound": "hbm" if f <= 64 else "mfma", "compulsory_bytes": b_comp,
                         "achieved_GBs": round(b_comp / t / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                         "frac": round(b_comp / t / 1e9 / HBM_PEAK_GBS, 4),
                         "l2_path_bytes": b_l2, "traffic": pmc_traffic([(f"k_mlp2r<{f},", "" if members == 1 else f"@{members}members"),
                                                 (f"k_mlp2<{f},", "" if members == 1 else f"@{members}members")]),
                         "bf16_tflops_issued": round(12 * f * f * e / t / 1e12, 1)}}


def hbm_leg(ga, mesh, graph, args, dev, f=None, m=None, what="c3 stack at c5's per-GPU load", order="auto"):
    """BASELINE c3's processor stack at c5's per-GPU member count: the regime where HBM bounds the path.
    ``order``: "auto" = the library's default precision (fp32-class: f16x3 in K8, bf16x6 in every other kernel),
    "auto_x6" = bf16x6 in every kernel, "auto_x3" = the bf16x3 split."""
    n, e = mesh.num_nodes, mesh.num_edges
    f, m, nl = f or args.hbm_channels, m or args.hbm_members, args.hbm_layers
    torch.manual_seed(23)
    layers = []
    for _ in range(nl):
        conv = ga.GCNConv(f, f).to(dev)
        with torch.no_grad():
            conv.bias.normal_(0.0, 0.1)
        layers.append((conv.lin.weight.detach(), conv.bias.detach(), True, order))
    plan = ga.StackForward(layers, graph)
    x = torch.stack([torch.randn(n, f, generator=torch.Generator().manual_seed(123 + k)) for k in range(m)]).to(dev)
    out = plan.run(x)
    prewarm(lambda: plan.run(x, out=out))
    samp = Sampler(ga, 2 * nl)
    steps = max(args.hbm_steps, (MIN_SAMPLES + nl - 1) // nl)
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(x, out=out, events=samp.new())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    summ, over = samp.summary(dev)
    dom = max(summ, key=lambda k: summ[k][1])
    cnt, tot = summ[dom]
    kind, fin, fout = dom
    avg = tot / cnt
    comp = compulsory_bytes(kind, n, e, fin, fout, m)
    ws_mib = 2 * 4 * m * n * f / 2 ** 20
    ns = {"auto": "f16" if kind == "wide" else 3, "auto_x6": 3, "auto_x3": 2}[order]
    tags = kernel_tags(kind, fin, fout, ns)
    launches_per_layer = 1
    if kind == "wide" and (fin, fout, ns) == (256, 256, 3):      # bf16x6 at 256 -> 256 = two 256 -> 128 launches
        tags, launches_per_layer = kernel_tags(kind, 256, 128, 3), 2
    # one member against the CPU oracle's first layer would take minutes at this width; parity at this size
    # is tests/test_gpu_wide.py::test_c3_layer_at_config_size_four_members
    return {
        "workload": f"{what}: {nl} chained GCN layers {f}->{f} + ReLU, {m} members, "
                    f"nu={args.nu} N={n} E={e}; {ws_mib:.0f} MiB in+out per layer (Infinity Cache: 256 MiB)",
        "members": m, "channels": f, "layers": nl, "steps": steps,
        "contraction": {"f16": "f16x3: two power-of-two-scaled fp16 images per operand, three MFMA terms -- fp32-class, "
                               "K8's split under the library default, ONE launch per layer",
                        3: "bf16x6: three bf16 images per operand, six MFMA terms -- fp32-class (precision \"bf16x6\")"
                           + ("; 256 -> 256 as two 256 -> 128 launches" if launches_per_layer == 2 else ""),
                        2: "bf16x3 (precision \"3xbf16\"): two bf16 images, three MFMA terms, ~17 bits per product"}[ns],
        "launches_per_layer": launches_per_layer,
        "ms_per_step": round(dt * 1e3, 4),
        "edges_per_s": round(m * nl * e / dt),
        "roofline": {"bound": "hbm", "kernel": f"{kind}_f32[{fin}->{fout}] x {m} members",
                     "achieved": round(comp / avg / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(comp / avg / 1e9 / HBM_PEAK_GBS, 4),
                     "compulsory_bytes_per_launch": comp, "avg_launch_us": round(avg * 1e6, 2),
                     "samples": cnt,
                     "traffic": (lambda v_: None if v_ is None else launches_per_layer * v_)(pmc_traffic(tags)),
                     "traffic_source": TRAFFIC_SOURCE,
                     "all_kernels_us": {f"{k[0]}[{k[1]}->{k[2]}]": round(v[1] / v[0] * 1e6, 2)
                                        for k, v in sorted(summ.items())}},
    }


def init_ranks(args):
    """(world, rank, dev) of this process; starts the ranks itself when nobody else did."""
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # BEFORE anything touches the GPU: this parent only starts N fresh interpreters and waits for them
        plan = launch_plan(args.gpus, sys.argv[1:])
        if args.launch_dry_run:
            print(json.dumps({"launcher": "bench.py self-launch (one fresh process per GPU)", "n_ranks": args.gpus,
                              "children": [{"argv": a, "env": e} for a, e in plan]}))
            raise SystemExit(0)
        raise SystemExit(run_plan(plan, timeout_s=args.rank_timeout))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_dry_run:
        print(json.dumps({"launcher": "none needed", "n_ranks": world, "children": []}))
        raise SystemExit(0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return world, rank, dev


def timed_allgather_object(ag0, ag1, final, world, dev):
    ag_ms = torch.tensor([ag0.elapsed_time(ag1)], dtype=torch.float64, device=dev)
    seen = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(ag_ms, op=dist.ReduceOp.MAX)
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    per_rank = final.numel() * 4
    return {"ms": round(float(ag_ms.item()), 4), "bytes_per_rank": per_rank,
            "busbw_GBs": round(per_rank * (world - 1) / (float(ag_ms.item()) * 1e-3) / 1e9, 2),
            "ranks_seen": int(seen.item()), "world_size": dist.get_world_size(),
            "backend": dist.get_backend(),
            "note": "one all_gather_into_tensor of the final states, inside the timed region; "
                    "busbw = bytes received per rank / time (each rank receives (world-1) shards)"}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            return next(ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name"))
    except (OSError, StopIteration):
        return "unknown CPU"


def cpu_thread_sweep(fn, budget_s, counts=None, warm=True):
    """Median time of fn() at each thread count (1, 8, 16, 32, 64, all -- those the host has); returns
    (best_threads, best_median, {threads: (median, reps)}, last result).  The stated baseline is the BEST
    configuration found, not the widest one (torch's index_add_ does not scale with threads)."""
    all_cores = torch.get_num_threads()
    counts = counts or sorted({c_ for c_ in (1, 8, 16, 32, 64, all_cores) if c_ <= all_cores})
    per_leg = budget_s / max(len(counts), 1)
    res, last = {}, None
    for th in counts:
        torch.set_num_threads(th)
        with torch.no_grad():
            tw = time.perf_counter(); last = fn(); one = time.perf_counter() - tw       # warm-up (or the only pass)
            reps = max(1, min(20, int(per_leg / max(one, 1e-3))))
            ts = [] if warm else [one]
            for _ in range(reps if warm else 0):
                tw = time.perf_counter(); last = fn(); ts.append(time.perf_counter() - tw)
        res[th] = (sorted(ts)[len(ts) // 2], reps)
    torch.set_num_threads(all_cores)
    best = min(res, key=lambda k_: res[k_][0])
    return best, res[best][0], res, last


def run_c5(args, world, rank, dev):
    """BASELINE configs[4]: the build-defined InteractionNet grid->mesh->grid forecaster (gwen_amd/forecaster.py;
    the reference has no such model -- SURVEY section 0 -- its only part kept is the eval's shape: compute locally,
    ONE all-gather at the end, /root/reference/src/gwen/models_gnn.py:471).  One STEP = one R-step autoregressive
    rollout of every local member (all of them through one launch set per step, captured once and replayed) + the
    single gather of the final states.  value = members * K / t  [members per second, whole job]."""
    import gwen_amd
    from gwen_amd import ensemble
    from gwen_amd.forecaster import InteractionForecaster, ensemble_forecast

    ch, nb, r_steps, m_local = args.c5_channels, args.c5_blocks, args.rollout_steps, args.c5_members_per_gpu
    members = m_local * world
    lo, hi = ensemble.member_range(members, rank, world)
    mesh = gwen_amd.geodesic_mesh(args.nu, reorder=None if args.reorder == "none" else args.reorder)
    n_grid, n_mesh = mesh.faces.shape[0], mesh.num_nodes
    torch.manual_seed(23)
    model = InteractionForecaster(ch, ch, nb).to(dev).eval()
    graphs = model.prepare(mesh, dev)
    x = torch.stack([torch.randn(n_grid, ch, generator=torch.Generator().manual_seed(23 + m))
                     for m in range(lo, hi)]).to(dev)
    cache = {}                                  # the captured step lives here: timed calls only replay

    def step():
        return ensemble_forecast(model, graphs, x, r_steps, members, graphed=True, batched=True, step_cache=cache)

    def barrier():
        if world > 1:
            dist.barrier()

    out = step()                                # capture + RCCL channel set-up, outside the timed region
    # steady-state run-in on local work only (a time-bounded loop must not contain a collective)
    prewarm(lambda: ensemble_forecast(model, graphs, x, 1, members, graphed=True, batched=True, step_cache=cache,
                                      gather=False), args.prewarm_ms / 1e3)
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    assert out.shape[0] == members and torch.isfinite(out).all()

    allgather = None
    if world > 1:               # the collective by itself (outside the timed region; the timed steps contain it)
        local = out[lo:hi].contiguous()
        ag0, ag1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ag0.record()
        ensemble.gather_members(local, members)
        ag1.record()
        torch.cuda.synchronize()
        allgather = timed_allgather_object(ag0, ag1, local, world, dev)

    edges = graphs.g2m.num_edges + nb * graphs.mesh.num_edges + graphs.m2g.num_edges
    line = {
        "metric": "ensemble members/s (autoregressive rollout incl. the final all-gather)",
        "value": members * args.steps / elapsed, "unit": "members/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c5: InteractionNet forecaster (build-defined; the reference has no grid/mesh model), "
                               f"geodesic mesh nu={args.nu}: grid {n_grid} cells -> mesh {n_mesh} vertices -> grid, "
                               f"{ch} channels, {nb} processor blocks, {r_steps}-step autoregressive rollout, "
                               f"{m_local} members/GPU batched through one launch set per step",
                   "grid_nodes": n_grid, "mesh_nodes": n_mesh, "edge_updates_per_model_step": edges,
                   "channels": ch, "processor_blocks": nb, "rollout_steps": r_steps, "members": members,
                   "node_order": args.reorder, "contraction": "3xbf16-split (fp32 storage and accumulation)",
                   "hip_graph": True, "rehearsal": "gloo, all ranks on GPU 0" if args.rehearse_gloo else None,
                   "parallelism": f"ensemble members sharded 1 rank = {m_local} members; one all-gather at end"},
        "edge_updates_per_s": members * r_steps * edges * args.steps / elapsed,
    }
    if allgather is not None:
        allgather["expected"] = allgather_expectation(allgather["bytes_per_rank"], world)
        line["allgather"] = allgather
    line["library"] = library_info(args)
    if rank == 0 and world == 1:
        # dominant kernel of the step: K6's edge kernel on the mesh->mesh blocks, by itself on the same batch
        side = edge_mlp_side_measurement(mesh, ch, dev, launches=10, members=m_local)
        t_k6 = side["us_per_launch"] * 1e-6
        e_b = side["edges"]
        if ch <= 64:
            comp = side["roofline"]["compulsory_bytes"]
            line["roofline"] = {"bound": "hbm", "kernel": f"k_mlp2r<{ch}> (K6 edge kernel) x {m_local} members",
                                "achieved": round(comp / t_k6 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(comp / t_k6 / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                "compulsory_bytes_per_launch": comp, "avg_launch_us": side["us_per_launch"]}
        else:
            flops = 12 * ch * ch * e_b             # two F x F contractions per edge, 3 bf16 terms each, 2 flop/MAC
            line["roofline"] = {"bound": "mfma", "kernel": f"K6 edge kernel at {ch} channels x {m_local} members",
                                "achieved": round(flops / t_k6 / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(flops / t_k6 / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                "traffic": None, "flops_per_launch_bf16_issued": flops,
                                "avg_launch_us": side["us_per_launch"],
                                "note": "bf16 MFMA flops ISSUED (3 terms per fp32-class product); fp32-equivalent "
                                        "work is a third of it"}
        if not args.no_cpu_baseline:
            from oracle import interaction_oracle as IO
            line["cpu_baseline"] = c5_cpu_baseline(IO, model, mesh, graphs, x, out, args, members, elapsed)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def c5_cpu_baseline(IO, model, mesh, graphs, x, out, args, members, elapsed):
    """The plain-torch restatement of the forecaster (oracle/interaction_oracle.py) on this host: ONE model step of
    ONE member (a bounded sample: a full 4-member x 4-step rollout at 256 channels is minutes of CPU work)."""
    import numpy as np
    from gwen_amd.forecaster import edge_features
    from gwen_amd.g2m import grid_mesh_edges
    g2m, m2g = grid_mesh_edges(mesh)
    cell = mesh.pos[mesh.faces].mean(axis=1)
    cell /= np.linalg.norm(cell, axis=1, keepdims=True)
    feats = [torch.from_numpy(v) for v in (edge_features(cell, mesh.pos, g2m),
                                           edge_features(mesh.pos, mesh.pos, mesh.edge_index),
                                           edge_features(mesh.pos, cell, m2g))]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    pos = torch.from_numpy(mesh.pos.astype(np.float32))
    eis = [torch.from_numpy(v) for v in (g2m, mesh.edge_index, m2g)]
    xc = x[0].cpu()
    all_c = torch.get_num_threads()
    with torch.no_grad():
        best_t, med, sweep, y1 = cpu_thread_sweep(
            lambda: IO.forecaster_step(sd, xc, pos, *eis, *feats, args.c5_blocks), args.cpu_seconds,
            counts=sorted({min(8, all_c), min(32, all_c)}), warm=False)   # one pass each: a step is ~10 s of CPU
    # one device step of the same member against it
    with torch.no_grad():
        got = model(x[0], graphs).cpu()
    err = float((got - y1).abs().max() / y1.abs().max())
    per_member = med * args.rollout_steps
    return {"value": 1.0 / per_member, "unit": "members/s", "cores": best_t, "kind": "port",
            "sample": f"{sweep[best_t][1]} single forecaster steps of one member (median {med*1e3:.0f} ms), "
                      f"x {args.rollout_steps} steps per rollout; torch {torch.__version__} CPU, {best_t} thread(s), best of "
                      f"a sweep on {cpu_model_name()}; oracle/interaction_oracle.py (build-defined model: parity unpinned)",
            "thread_sweep": {str(th): {"median_ms": round(v[0] * 1e3, 1), "passes": v[1]} for th, v in sorted(sweep.items())},
            "gpu_vs_oracle_rel_err_one_step": err}


def library_info(args):
    """Which libgwen_hip.so this line was measured on (path + source digest).  A variant build (GWEN_HIP_LIB: the
    experiment / ablation builds of tools/experiments) is refused unless --allow-variant says the caller knows."""
    from gwen_amd import _lib
    if _lib.is_variant() and not args.allow_variant:
        raise SystemExit(f"GWEN_HIP_LIB={os.environ.get('GWEN_HIP_LIB')} replaces the product library: unset it, or "
                         f"pass --allow-variant (the line then says library.variant = true)")
    return _lib.library_stamp()


def allgather_expectation(bytes_per_rank, world):
    """What the one collective should cost on xGMI (point-to-point, 7 links x ~153 GB/s per GPU per the guide): every
    rank receives (world - 1) shards, in the best case each over its own link."""
    recv = bytes_per_rank * max(world - 1, 0)
    links = min(max(world - 1, 1), 7)
    return {"bytes_per_rank": bytes_per_rank, "bytes_received_per_rank": recv,
            "expected_us_direct_links": round(bytes_per_rank / 153e9 * 1e6 * max(world - 1, 0) / links, 1),
            "expected_us_one_link_ring": round(recv / 153e9 * 1e6, 1),
            "note": "xGMI ~153 GB/s per link and direction, 7 links per GPU (MI355X_MICROARCH.md): between these two "
                    "figures is healthy; RCCL adds ~20-50 us of launch / protocol latency at small sizes"}


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if "WORLD_SIZE" in os.environ or args.gpus == 1:
        library_info(args)                               # refuse a variant library before any work
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.workload == "c2" and not args.no_live_traffic \
            and not (args.no_hbm_leg and args.no_edge_mlp) and "ROCPROF_OUTPUT_PATH" not in os.environ:
        collect_live_traffic(args)                       # (not when this run is itself under rocprofv3)
    world, rank, dev = init_ranks(args)
    if args.workload == "c5":
        return run_c5(args, world, rank, dev)

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

    def set_order(which):
        for mod in model.modules():
            if isinstance(mod, gwen_amd.GCNConv):
                if which == "unfused":
                    mod.order = "aggregate_first" if mod.in_channels < mod.out_channels else "transform_first"
                elif which == "fused_exact":
                    mod.order = "fused_exact" if gwen_amd.ops.layer_supported(mod.in_channels, mod.out_channels) else \
                        ("aggregate_first" if mod.in_channels < mod.out_channels else "transform_first")
                elif which == "bf16x3":
                    mod.order = "auto_x3"
                else:
                    mod.order = "auto"
    set_order(args.order)
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
    samp = Sampler(gwen_amd, 2 * layers)
    n_sets = (args.steps + stride - 1) // stride
    ev_sets = [samp.new() for _ in range(n_sets)]

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

    prewarm(step, args.prewarm_ms / 1e3)
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
    if world > 1:
        ag0, ag1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ag0.record()
        gathered = ensemble.gather_members(final, members)
        ag1.record()
    else:
        gathered = final
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    assert gathered.shape[0] == members and torch.isfinite(gathered).all()

    allgather = timed_allgather_object(ag0, ag1, final, world, dev) if world > 1 else None

    # ---- more instrumented steps (outside the timed region) until every kernel has MIN_SAMPLES ------
    while samp.min_samples() < MIN_SAMPLES:
        step(samp.new())
        torch.cuda.synchronize()
    summ, ev_over = samp.summary(dev)

    # ---- roofline of the dominant kernel of the c2 step: the L2 is the level that binds ---------------
    dom_key = max(summ, key=lambda k: summ[k][1])
    launches, total_s = summ[dom_key]
    kind, fin, fout = dom_key
    b_l2 = algorithmic_bytes(kind, n, e, fin, fout) * m_local
    b_comp = compulsory_bytes(kind, n, e, fin, fout, m_local)
    avg_raw = total_s / launches
    # The brackets over-read (the first kernel of a step by ~2 us: its bracket also holds the launch gap behind the step
    # before): the kernels of a step can not take longer than the step, so when the per-kernel means sum to more than the
    # measured step they are scaled down together (bracket_scale < 1; rocprofv3's kernel-only averages agree with the
    # scaled figures, profiles/)
    ksum = sum(v[1] / v[0] for v in summ.values())
    bracket_scale = min(1.0, (elapsed / args.steps) / ksum) if ksum > 0 else 1.0
    avg_s = avg_raw * bracket_scale
    achieved = b_l2 / avg_s / 1e9
    tag = kernel_tags(kind, fin, fout, {"auto": 3, "unfused": 0, "bf16x3": 2, "fused_exact": 0}[args.order])
    default_c2 = (n, e, c, h, m_local) == (100002, 600000, 64, 64, 1)
    per_layer_us = {f"{k[0]}[{k[1]}->{k[2]}]": round(v[1] / v[0] * bracket_scale * 1e6, 2) for k, v in sorted(summ.items())}
    roofline = {
        "bound": "l2", "kernel": f"{kind}_f32[{fin}->{fout}]", "achieved": round(achieved, 1),
        "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / L2_PEAK_GBS, 4),
        "traffic": pmc_traffic(tag) if default_c2 else None,
        "traffic_source": TRAFFIC_SOURCE,
        "l2_path_bytes_per_launch": b_l2, "frac_of_l2_gather_rate": round(achieved / L2_GATHER_GBS, 4),
        "compulsory_bytes_per_launch": b_comp,
        "frac_hbm_compulsory": round(b_comp / avg_s / 1e9 / HBM_PEAK_GBS, 4),
        "avg_launch_us": round(avg_s * 1e6, 2), "samples": launches,
        "avg_launch_us_raw_bracket": round(avg_raw * 1e6, 2), "bracket_scale": round(bracket_scale, 4),
        "kernel_sum_us": round(ksum * bracket_scale * 1e6, 2),
        "event_record_overhead_us": round(ev_over * 1e6, 2), "all_kernels_us": per_layer_us,
        "why_l2": "the c2 working set (<= 3 x 25.6 MB) never leaves the L2s / Infinity Cache, so HBM does not "
                  "bound this kernel; the HBM-bound measurement of the same path is hbm_leg",
    }

    value = members * layers * e * args.steps / elapsed
    # one 64 -> 64 message+aggregate pass by itself (the last layer: a whole-layer kernel at 64 channels)
    last_key = [k for k in summ if k[1] == h and k[2] == c and k[0] in ("layer", "wide")]
    pass64 = None
    if last_key:
        cnt_, tot_ = summ[last_key[0]]
        pass64 = round(m_local * e / (tot_ / cnt_))
    line = {
        "metric": "mesh edges/s (message+aggregate)", "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c2: geodesic mesh nu={args.nu} N={n} E={e}, GNNModel forward "
                               f"C={c} H={h} (6 GCNConv layers at widths "
                               f"{'/'.join(str(min(a, b)) for a, b in widths)} gathered), {m_local} member/GPU",
                   "nodes": n, "edges": e, "channels": c, "hidden": h, "layers": layers,
                   "members": members, "node_order": args.reorder, "kernel_order": args.order,
                   "contraction": {"auto": "the library default, fp32-class on each kernel's own split -- at this "
                                           "model's widths (<= 64) bf16x6: three bf16 images per operand, six MFMA "
                                           "terms, fp32 accumulation (see gpu_vs_oracle_rel_err; siblings: bf16x3, "
                                           "exact_f32; K8 runs f16x3: hbm_leg)",
                                   "bf16x3": "bf16x3 split: two bf16 images per operand, three MFMA terms",
                                   "fused_exact": "fp32-input MFMA (exact fp32 products)",
                                   "unfused": "fp32-input MFMA in K3 (explicit orders contract in fp32)"}[args.order],
                   "hip_graph": bool(args.graph), "rehearsal": "gloo, all ranks on GPU 0" if args.rehearse_gloo else None,
                   "parallelism": f"ensemble members sharded 1 rank = {m_local} member(s); one all-gather at end"},
        "members_per_s": members * args.steps / elapsed,
        "edges_per_s_64ch_pass": pass64,
        "roofline": roofline,
    }
    if allgather is not None:
        allgather["expected"] = allgather_expectation(allgather["bytes_per_rank"], world)
        line["allgather"] = allgather
    line["library"] = library_info(args)

    single = rank == 0 and world == 1
    # ---- siblings of the headline on the other contractions (outside the timed region) ----------------------
    def sibling(which, text):
        set_order(which)
        plan_x = gwen_amd.StackForward(model.stack(), graph)
        out_x = plan_x.run(x)
        prewarm(lambda: plan_x.run(x, out=out_x))
        k_x = max(20, min(args.steps, 100))
        tx = time.perf_counter()
        for _ in range(k_x):
            plan_x.run(x, out=out_x)
        torch.cuda.synchronize()
        dtx = (time.perf_counter() - tx) / k_x
        set_order(args.order)
        return {"ms_per_step": round(dtx * 1e3, 5), "edges_per_s": round(members * layers * e / dtx),
                "contraction": text, "steps": k_x,
                "max_rel_diff_vs_headline": float((out_x - out).abs().max() / out_x.abs().max())}

    if single and not args.no_exact and args.order == "auto":
        line["exact_f32"] = sibling("fused_exact", "fp32-input MFMA (v_mfma_f32_16x16x4_f32), order fused_exact")
        line["bf16x3"] = sibling("bf16x3", "two bf16 images per operand, three MFMA terms (precision \"3xbf16\"): "
                                           "7e-6 relative on this model, inside the 1e-4 contract")
        # comparable round over round: rounds 1-2 quoted `value` on this tier (BENCH_r02: 41.46 G); since round 3 `value`
        # is the fp32-class default, ~5 % slower on this model
        line["value_tier_3xbf16"] = line["bf16x3"]["edges_per_s"]

    # ---- HBM-bound leg ----------------------------------------------------------------------------------
    if single and not args.no_hbm_leg:
        # 256 channels, the LIBRARY DEFAULT (fp32-class): K8 on the scaled fp16 split, one launch per layer.  Beside it
        # the faster 17-bit tier (precision "3xbf16") and what the default was before f16x3 (bf16x6: W's three images
        # for 256 output columns exceed the registers, so two 256 -> 128 launches per layer)
        line["hbm_leg"] = hbm_leg(gwen_amd, mesh, graph, args, dev, order="auto")
        line["hbm_leg"]["tier_3xbf16"] = hbm_leg(gwen_amd, mesh, graph, args, dev, order="auto_x3")
        line["hbm_leg"]["precision_bf16x6_two_launches"] = hbm_leg(gwen_amd, mesh, graph, args, dev, order="auto_x6")
        # the same layer kernel at c2's 64 channels with enough members to leave the Infinity Cache (default: f16x3 on K8's
        # two-chunk pipeline), the 3xbf16 tier and precision "bf16x6" beside it
        line["hbm_leg_64ch"] = hbm_leg(gwen_amd, mesh, graph, args, dev, f=args.channels, m=args.hbm_members_narrow,
                                       what="c2's width beyond the Infinity Cache", order="auto")
        line["hbm_leg_64ch"]["tier_3xbf16"] = hbm_leg(
            gwen_amd, mesh, graph, args, dev, f=args.channels, m=args.hbm_members_narrow,
            what="c2's width beyond the Infinity Cache", order="auto_x3")
        line["hbm_leg_64ch"]["precision_bf16x6"] = hbm_leg(
            gwen_amd, mesh, graph, args, dev, f=args.channels, m=args.hbm_members_narrow,
            what="c2's width beyond the Infinity Cache", order="auto_x6")

    # ---- side measurement (outside the timed region, N = 1 only): the InteractionNet edge-MLP kernel
    # K6 on the same mesh at the same width -- the block BASELINE.json's north_star names; the headline
    # value above stays the reference's own layer (GCNConv) -------------------------------------------
    if single and not args.no_edge_mlp and h in (32, 64, 128, 256):
        if not args.edge_mlp_skip_single:
            line["edge_mlp_block"] = edge_mlp_side_measurement(mesh, h, dev)
        # ... and on 8 members at once (2.9 GB of compulsory traffic at 64 channels: nothing stays in a cache)
        if args.edge_mlp_members > 1:
            line[f"edge_mlp_block_{args.edge_mlp_members}_members"] = edge_mlp_side_measurement(
                mesh, h, dev, launches=10, members=args.edge_mlp_members)

    # ---- CPU baseline: the torch oracle on this host's cores (rank 0, N = 1 only) ----------------
    if single and not args.no_cpu_baseline:
        from oracle import gcn_oracle as O
        ref = O.OracleGNNModel(O.OracleGNNConfig(n, n, c, c, h))
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=True)
        xc = (x if x.dim() == 2 else x[0]).cpu()
        eic = torch.from_numpy(mesh.edge_index)
        cpu_model = cpu_model_name()
        all_cores = torch.get_num_threads()
        with torch.no_grad():
            best_t, med, sweep, yc = cpu_thread_sweep(lambda: ref(xc, eic), 2.0 * args.cpu_seconds)
        got = (out if out.dim() == 2 else out[0]).cpu()
        err = float((got - yc).abs().max() / yc.abs().max())
        line["cpu_baseline"] = {
            "value": layers * e / med, "unit": "edges/s", "cores": best_t, "kind": "port",
            "sample": f"{sweep[best_t][1]} full GNNModel.forward passes of the same c2 workload (1 member), median "
                      f"{med*1e3:.1f} ms, torch {torch.__version__} CPU, {best_t} thread(s) -- the best of a sweep -- "
                      f"on {cpu_model} ({all_cores} hardware threads), oracle/gcn_oracle.py",
            "thread_sweep": {str(th): {"edges_per_s": round(layers * e / v[0]), "median_ms": round(v[0] * 1e3, 1),
                                       "passes": v[1]} for th, v in sorted(sweep.items())},
            "gpu_vs_oracle_rel_err": err,
        }
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
