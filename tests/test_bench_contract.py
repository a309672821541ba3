"""bench.py: the byte models behind its roofline objects (CPU) and the JSON line's contract (GPU, tiny run)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_byte_models_match_the_survey_formulas():
    import bench
    n, e = 100_002, 600_000                                       # c2: nu = 100 mesh, directed edges without loops
    # SURVEY 8(d) for a gather at width F: F * 4 * (E' + N) with E' = E + N stored entries, + indices
    assert bench.algorithmic_bytes("propagate", n, e, 64, 64) == 4 * 64 * (e + 2 * n) + 8 * e + 8 * n
    # the c2 dominant kernel (K5 64 -> 64 -> 32, stored 32 wide): the figures DESIGN.md section 5 quotes
    assert bench.algorithmic_bytes("chain", n, e, 64, 32) == 197_600_784
    assert bench.compulsory_bytes("chain", n, e, 64, 32) == 44_408_984
    # compulsory = every input, output, index and weight byte once; members scale the activations only
    one = bench.compulsory_bytes("wide", n, e, 256, 256, 1)
    four = bench.compulsory_bytes("wide", n, e, 256, 256, 4)
    assert four - one == 3 * 4 * n * (256 + 256)
    assert bench.compulsory_bytes("linear", n, e, 256, 768) == 4 * n * (256 + 768) + 4 * 256 * 768
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.MIN_SAMPLES >= 10


def test_default_flags_are_the_contract(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.steps > 0, a.warmup > 0) == (1, True, True)
    assert a.nu == 100 and a.channels == 64 and a.hidden == 64    # BASELINE.json configs[1]
    assert a.prewarm_ms > 0


@pytest.mark.gpu
def test_bench_line_contract_on_a_tiny_run(hip_lib):
    """One JSON line with the fields the driver parses, `roofline` (frac <= 1, >= 10 samples) and `cpu_baseline`."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "2", "--no-hbm-leg", "--no-exact",
           "--cpu-seconds", "1", "--edge-mlp-members", "0", "--prewarm-ms", "20"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("l2", "hbm", "mfma") and 0.0 < r["frac"] <= 1.0 and r["samples"] >= 10
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["gpu_vs_oracle_rel_err"] <= 1e-4
    assert d["value"] > 1e9 and abs(d["value"] - 6 * 600_000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_self_launch_plan_dry_run():
    """`python bench.py --gpus N` without a launcher starts its own N ranks (the reference's mp.spawn,
    /root/reference/src/gwen/train_gnn.py:144-152); --launch-dry-run prints the plan and touches no GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
                          "--launch-dry-run"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    kids = plan["children"]
    assert plan["n_ranks"] == 4 and len(kids) == 4
    assert [k["env"]["RANK"] for k in kids] == ["0", "1", "2", "3"]
    assert [k["env"]["LOCAL_RANK"] for k in kids] == ["0", "1", "2", "3"]
    assert {k["env"]["WORLD_SIZE"] for k in kids} == {"4"} and {k["env"]["MASTER_ADDR"] for k in kids} == {"127.0.0.1"}
    assert len({k["env"]["MASTER_PORT"] for k in kids}) == 1 and {k["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] for k in kids} == {"0"}
    for k in kids:
        assert k["argv"][1].endswith("bench.py") and "--launch-dry-run" not in k["argv"]
        assert k["argv"][2:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    # under a launcher (WORLD_SIZE set) nothing is started
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-dry-run"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env={**env, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["children"] == []


def test_self_launch_runs_children_and_propagates_failure():
    import bench
    ok = [([sys.executable, "-c", "import os; assert os.environ['WORLD_SIZE'] == '3'"], {"RANK": str(r), "WORLD_SIZE": "3"})
          for r in range(3)]
    assert bench.run_plan(ok, poll_s=0.05) == 0
    # rank 1 fails with code 7: the launcher returns it and ends the rank that would run for a minute
    bad = [([sys.executable, "-c", "import os, sys, time; r = int(os.environ['RANK']); "
                                   "time.sleep(60) if r == 0 else sys.exit(7 if r == 1 else 0)"], {"RANK": str(r)})
           for r in range(3)]
    import time
    t0 = time.perf_counter()
    assert bench.run_plan(bad, poll_s=0.05) == 7
    assert time.perf_counter() - t0 < 30


def test_self_launch_kills_a_rank_that_hangs():
    """A rank stuck in a collective must not hang the parent: past the wall-clock limit the ranks are terminated, one
    that ignores SIGTERM is killed after the grace period, and the result is non-zero."""
    import bench
    import time
    hang = [([sys.executable, "-c", "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(120)"],
             {"RANK": "0"}),
            ([sys.executable, "-c", "import time; time.sleep(120)"], {"RANK": "1"})]
    t0 = time.perf_counter()
    assert bench.run_plan(hang, poll_s=0.05, timeout_s=1.0, grace_s=1.0) == 124
    assert time.perf_counter() - t0 < 30


def test_variant_library_is_refused(tmp_path):
    """GWEN_HIP_LIB (an experimental / ablated build) must not produce a bench line or a green test run unasked."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["GWEN_HIP_LIB"] = str(tmp_path / "libgwen_hip.other.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "GWEN_HIP_LIB" in (out.stderr + out.stdout)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi.py"), "-q", "-x"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "GWEN_HIP_LIB" in (out.stderr + out.stdout)


def test_c5_workload_defaults(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "c5"])
    a = bench.parse()
    # BASELINE configs[4]: 256 channels, 4-step rollout, 32 members over 8 GPUs = 4 per GPU
    assert (a.c5_channels, a.rollout_steps, a.c5_members_per_gpu, a.c5_blocks) == (256, 4, 4, 4)
    assert a.steps == 10 and a.warmup == 2


@pytest.mark.gpu
def test_c5_bench_line_on_a_small_run(hip_lib):
    """--workload c5 at 64 channels: members/s line with the roofline of K6 and the oracle-checked CPU baseline."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--c5-channels", "64", "--steps", "2",
           "--warmup", "1", "--cpu-seconds", "2", "--prewarm-ms", "20"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["unit"] == "members/s" and d["n_gpus"] == 1 and d["config"]["members"] == 4
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] <= 1
    assert d["cpu_baseline"]["gpu_vs_oracle_rel_err_one_step"] <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c2", "c5"])
def test_two_ranks_self_launched_rehearsal(hip_lib, workload):
    """`python bench.py --gpus 2` with NO launcher: the parent starts both ranks itself and the N > 1 path runs end to
    end -- member sharding, MAX-reduced timing, the one gather inside the timed region, the `allgather` object.  A
    one-GPU box cannot give RCCL two devices, so this is the rehearsal mode (both ranks on GPU 0, gloo, the gather
    through the host): it proves the code path the driver's 2/4/8-GPU runs take, not a number."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-gloo", "--steps", "3", "--warmup", "1",
           "--prewarm-ms", "10"]
    if workload == "c5":
        cmd += ["--workload", "c5", "--c5-channels", "64", "--c5-members-per-gpu", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rehearsal"]
    ag = d["allgather"]
    assert ag["ranks_seen"] == 2 and ag["world_size"] == 2 and ag["backend"] == "gloo" and ag["bytes_per_rank"] > 0
    assert d["config"]["members"] == (2 if workload == "c2" else 4)
    assert d["value"] > 0 and "cpu_baseline" not in d and "hbm_leg" not in d


@pytest.mark.gpu
def test_default_line_carries_live_pmc_traffic_and_the_library_stamp():
    """One short default-shaped run (N = 1): the contract keys, the library stamp, and `roofline.traffic` collected by
    this very run (two rocprofv3 --pmc child passes) -- within 1.0 .. 1.25 x the compulsory bytes for the HBM-bound legs
    (wasted re-reads would show here first); `ag["expected"]` on the N > 1 line is covered by the rehearsal test."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--hbm-steps", "3",
           "--no-cpu-baseline", "--no-exact"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "library"):
        assert key in d, key
    assert d["steps"] == 20 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["library"]["variant"] is False and d["library"]["matches_sources"] is True
    r = d["roofline"]
    assert r["kernel_sum_us"] <= d["ms_per_step"] * 1e3 * 1.001 and 0 < r["bracket_scale"] <= 1.0
    assert r["traffic_source"].startswith("LIVE"), r["traffic_source"]
    leg = d["hbm_leg"]
    assert leg["launches_per_layer"] == 1 and "f16x3" in leg["contraction"]
    for lg in (leg, leg["tier_3xbf16"], d["hbm_leg_64ch"], d["hbm_leg_64ch"]["tier_3xbf16"]):
        rr = lg["roofline"]
        assert rr["traffic"] is not None and 1.0 <= rr["traffic"] / rr["compulsory_bytes_per_launch"] <= 1.25, rr
    two = leg["precision_bf16x6_two_launches"]
    assert two["launches_per_layer"] == 2 and two["roofline"]["traffic"] / two["roofline"]["compulsory_bytes_per_launch"] > 1.4
    e8 = d["edge_mlp_block_8_members"]["roofline"]
    assert e8["traffic"] is not None and 1.0 <= e8["traffic"] / e8["compulsory_bytes"] <= 1.25
