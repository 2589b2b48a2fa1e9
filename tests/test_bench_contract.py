"""bench.py contract: on a machine without a GPU it refuses to run (there is no CPU path to time); on the GPU it
prints exactly one JSON line carrying the driver's keys, BASELINE.json's metric, and the roofline / cpu_baseline
objects.  `--gpus N` from a plain shell starts N ranks itself; under torch.distributed.run it is rank code."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    """An environment without any launcher variables: what the driver's plain `python bench.py --gpus N` sees."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_launcher_starts_n_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns 2 rank processes which rendezvous (gloo here;
    --dry-run because this container has no GPU and there is no CPU compute path) and rank 0 prints ONE line with
    n_gpus == 2."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["steps"] == 2


def test_rank_code_under_torch_distributed_run():
    """The driver's N>1 form: torch.distributed.run starts the ranks, bench.py must not spawn again."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH, "--gpus", "2", "--dry-run",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, cwd=ROOT, env=clean_env(),
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["dry_run"] is True


def test_launcher_fails_when_a_rank_fails():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, cwd=ROOT, env=clean_env(), timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
    assert not r.stdout.strip()


def test_bench_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=clean_env())
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--ramp-seconds",
                        "0.3", "--batch", "8192", "--c4-batch", "512"], capture_output=True, text=True, cwd=ROOT,
                       env=clean_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = one_json_line(r.stdout)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] == base["metric"] and d["unit"] == "NTT/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    # frac is priced against the HBM roofline on the bytes the kernel MOVES; the limiter named is the real one
    assert rf["bound"] == "valu/power" and rf["priced_against"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "NTT/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # value = units processed / time: 2 transforms per polynomial per step
    assert abs(d["value"] - 2 * 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the other BASELINE configs ride on the same line
    names = [c.get("config", "") for c in d["configs"]]
    assert any(n.startswith("C3") for n in names) and any(n.startswith("C5") for n in names)
    assert any(n.startswith("C4") for n in names)
    for c in d["configs"]:
        assert "error" not in c, c
    # the fused kernel moves 24 KiB per polynomial; the per-transform accounting (2 x 16 KiB) is a separate field
    assert abs(rf["achieved"] - 3 * 1024 * 8 * 8192 / (rf["avg_launch_ms"] * 1e-3) / 1e9) / rf["achieved"] < 1e-9
    assert abs(rf["per_transform_frac"] / rf["frac"] - 4.0 / 3.0) < 1e-9
    assert rf["traffic"] is None and d["fwd_inv_standalone_value"] > 0   # no PMC figure at this batch
    assert rf["csrc_hash"] and len(rf["csrc_hash"]) == 16
    # the correctness gate rides on the line: fused and three-launch step against the oracle on a fresh batch
    assert d["verified"] is True and d["verification"]["mismatches"] == [] and "[0, 4097, 8191]" in d["verification"]["checker"]


@pytest.mark.gpu
def test_bench_gate_closes_on_a_wrong_word():
    """The correctness gate of the printed number: with one bit of the verified batch flipped the line carries no value, says why, and the
    command fails (status 4) -- SURVEY 8(d) / BASELINE.md 3 "correctness gate before any timing is accepted"."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--ramp-seconds", "0.2", "--batch", "8192", "--no-extra",
                        "--no-cpu-baseline", "--selftest-corrupt"], capture_output=True, text=True, cwd=ROOT, env=clean_env(), timeout=600)
    assert r.returncode == 4, (r.returncode, r.stderr[-1500:])
    d = one_json_line(r.stdout)
    assert d["value"] is None and d["verified"] is False and d["verification"]["ran"] is True
    assert any("polynomial 8191" in m for m in d["verification"]["mismatches"]), d["verification"]


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_ranks_on_the_gpu_box(launcher):
    """world_size 2 through both launch forms; gloo so that both ranks may share the box's single GPU.  Checks the
    multi-rank line (n_gpus, whole-job value) and the C4 legs: shard-resident and scatter -> fwd -> gather."""
    tail = ["--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--ramp-seconds", "0.2", "--batch",
            "4096", "--c4-batch", "64", "--no-cpu-baseline"]
    if launcher == "self":
        cmd = [sys.executable, BENCH] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()), BENCH] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=clean_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "dry_run" not in d
    assert abs(d["value"] - 2 * 2 * 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert abs(d["per_gpu_value"] * 2 - d["value"]) / d["value"] < 1e-9
    by_name = {c["config"]: c for c in d["configs"]}
    assert "error" not in json.dumps(d["configs"]), d["configs"]
    e2e = by_name["C4 prime64 N=16384 end to end"]
    assert e2e["n_gpus"] == 2 and e2e["scatter_s"] > 0 and e2e["compute_s"] > 0 and e2e["gather_s"] > 0
    assert by_name["C4 prime64 N=16384 shard-resident"]["n_gpus"] == 2


@pytest.mark.gpu
def test_one_rank_rccl_rehearsal():
    """The driver's N>1 form with the `nccl` (= RCCL) backend, at the one rank this box can host: RCCL loads, the process
    group comes up on the device, the timing all-reduces run on device tensors, and the C4 end-to-end leg drives
    shard.scatter_batch / gather_batch on device tensors through RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), BENCH, "--gpus", "1", "--dist-backend", "nccl", "--force-dist", "--steps", "2",
           "--warmup", "1", "--ramp-seconds", "0.2", "--batch", "4096", "--c4-batch", "256", "--no-cpu-baseline"]
    env = clean_env()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["rccl_ranks"] == 1 and d["config"]["dist_backend"] == "nccl"
    by_name = {c["config"]: c for c in d["configs"]}
    assert "error" not in json.dumps(d["configs"]), d["configs"]
    e2e = by_name["C4 prime64 N=16384 end to end"]
    assert e2e["scatter_s"] > 0 and e2e["compute_s"] > 0 and e2e["gather_s"] > 0


@pytest.mark.gpu
def test_line_carries_clock_and_power_and_the_watchdog_keeps_the_line():
    """(1) roofline.sclk_mhz / power_w: sampled from the card's hwmon files (or amdsmi) while the timed region ran, so that a slow
    driver run can be told from a slow build.  (2) A leg after the timed region that does not return must not cost the headline
    line: the watchdog prints it with what finished, names the running leg, flags the line (`extras_watchdog`) and ends the rank
    -- with status 0 by default (a launcher must not discard a complete headline), 3 with --extras-strict."""
    base = [sys.executable, BENCH, "--steps", "50", "--warmup", "2", "--ramp-seconds", "0.5", "--batch", "16384", "--c4-batch",
            "256", "--no-cpu-baseline"]
    r = subprocess.run(base, capture_output=True, text=True, cwd=ROOT, env=clean_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = one_json_line(r.stdout)
    rf = d["roofline"]
    assert rf["sensors"]["source"], rf["sensors"]
    assert 400 <= rf["sclk_mhz"] <= 2500 and 100 <= rf["power_w"] <= 1500, (rf["sclk_mhz"], rf["power_w"])
    assert rf["sensors"]["fused_kernel"]["samples"] >= 1 and rf["sensors"]["fwd_kernel"]["samples"] >= 1
    c3 = [c for c in d["configs"] if c["config"].startswith("C3")][0]
    assert c3["sclk_mhz"] and c3["power_w"]
    # the watchdog: one second is not enough for the extras
    assert d["extras_watchdog"] is False
    for strict, want in ((False, 0), (True, 3)):
        r = subprocess.run(base + ["--extras-timeout", "1"] + (["--extras-strict"] if strict else []), capture_output=True,
                           text=True, cwd=ROOT, env=clean_env(), timeout=900)
        assert r.returncode == want, (r.returncode, r.stderr[-2000:])
        d = one_json_line(r.stdout)
        assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["extras_watchdog"] is True
    last = d["configs"][-1]
    assert "watchdog" in last["error"] and last["running_phase"] and last["seconds_in_phase"] is not None
