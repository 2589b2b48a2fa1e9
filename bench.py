#!/usr/bin/env python3
"""Headline benchmark: forward+inverse NTTs/sec, prime64, N=1024, batched (BASELINE.json).

One step = one pass of the hot path over one device-resident batch (configs[1]):
    fwd(A)  ->  mul_assign_normalize(A, B^)  ->  inv(A)        A: 65536 polynomials x 1024 x u64 (512 MiB)
i.e. a negacyclic product against a pre-transformed operand: 2 transforms per polynomial per step
plus the pointwise pass (which is inside the timed region and earns no units).  Inputs are synthetic
(splitmix64, generated on the device) and resident in HBM before the timed region starts.

N GPUs: one process per GPU, each owns its own shard of `batch` polynomials (independent units, no
collective on the data path) -> weak scaling.  Timing: barrier + synchronize on both sides of exactly
K steps, max over ranks, rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P62 = 4611686018427322369   # benches/ntt.rs:115  largest prime = 1 mod 2^16 below 2^62
N = 1024
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cpus():
    """CPUs this process may actually run on: affinity mask, capped by the cgroup CPU quota (a GPU box hands each
    job a share of the host, e.g. 16 of 256 hardware threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    period = int(g.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(n, p, seconds_budget=12.0):
    """The oracle (a C restatement of the reference's CPU path; the Rust crate itself cannot be built in this
    image) timed on this host's cores on a bounded sample of the same workload.  Where the host has AVX-512F+DQ the
    restatement of the reference's AVX-512 engine (src/prime64/shoup.rs:10-156) is what `value` reports -- it is the
    fastest path the reference would take on this CPU -- with the scalar engine's rates beside it."""
    import numpy as np  # noqa: F401
    from oracle import pyoracle
    try:
        pyoracle.build(native=True)   # -march=native for the machine that does the timing
        native = True
    except Exception:
        native = False
    plan = pyoracle.Plan.try_new(n, p, 64, native=native)
    cores = usable_cpus()
    sample = 16384
    buf = pyoracle.fill_uniform(sample * n, p, 0x5EED0002, 64)

    def rates(avx512, budget):
        # single thread (the reference's criterion harness is single-threaded: benches/ntt.rs:94-105)
        single = 2 * 2048 / plan.fwd_inv_loop(buf[: 2048 * n], 1, 1, avx512)
        # all host threads: every thread loops over its share long enough that thread start-up does not matter
        per_thread = max(1, sample // cores)
        reps = max(1, int(budget * 0.5 * single / (2 * per_thread)))     # aim at ~budget/2 seconds per call
        reps = min(reps, 4096)
        calls, tot = 0, 0.0
        t0 = time.perf_counter()
        while calls < 2 or (time.perf_counter() - t0 < budget and calls < 8):
            tot += plan.fwd_inv_loop(buf, cores, reps, avx512)
            calls += 1
        return single, 2.0 * reps * sample * calls / tot, reps * calls

    avx = native and plan.avx512_available()
    s_single, s_multi, s_reps = rates(False, seconds_budget / (4 if avx else 2))
    out = {"unit": "NTT/s", "cores": cores, "kind": "port", "scalar_value": s_multi, "scalar_single_thread_value": s_single}
    if avx:
        v_single, v_multi, v_reps = rates(True, seconds_budget / 4)
        out.update(value=v_multi, single_thread_value=v_single, isa="avx512f+dq")
        reps, single = v_reps, v_single
    else:
        out.update(value=s_multi, single_thread_value=s_single, isa="scalar")
        reps, single = s_reps, s_single
    out["sample"] = ("%d polynomials x (fwd+inv), N=%d, p=%d, %d repetitions inside each of %d threads (%s engine); single-thread "
                     "rate %.0f NTT/s; scalar engine: %.0f NTT/s on all threads, %.0f single; %s build of "
                     "oracle/cntt_oracle.c" % (sample, n, p, reps, cores, out["isa"], single, s_multi, s_single,
                                                "-O3 -march=native" if native else "-O3 portable"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="polynomials per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one rank per GPU) is what the driver uses; gloo lets several ranks share one "
                         "GPU to rehearse the multi-process path on a 1-GPU box")
    ap.add_argument("--ramp-seconds", type=float, default=2.0,
                    help="untimed back-to-back steps before the W warm-up steps, so that the timed region "
                         "runs at the steady-state DVFS clock instead of inside the ramp from idle")
    args = ap.parse_args()

    import torch
    import concrete_ntt_amd as cntt
    from concrete_ntt_amd import prime64

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(ndev, 1)   # rehearsal mode: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    batch = args.batch
    plan = prime64.Plan.try_new(N, P62)
    dev = torch.device("cuda", local_rank)
    a = torch.empty(batch * N, dtype=torch.int64, device=dev)
    b = torch.empty(batch * N, dtype=torch.int64, device=dev)
    # every rank owns a different shard of the synthetic stream
    cntt.fill_uniform(a, P62, 0x5EED0002 + rank * batch * N)
    cntt.fill_uniform(b, P62, 0x5EED1002 + rank * batch * N)
    plan.fwd_batch(b)   # B^ : the pre-transformed operand

    def step_unfused():
        plan.fwd_batch(a)
        plan.mul_assign_normalize_batch(a, b)
        plan.inv_batch(a)

    def step():  # same values as step_unfused, one fused kernel (fwd -> pointwise -> inv in registers/LDS)
        plan.mul_ntt_batch(a, b)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # device clock ramp (untimed): the GPU idles at ~100 MHz and the package power controller needs
    # hundreds of milliseconds of sustained load to settle (profiles/r01_power_clock_lab.txt)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same step as three API-faithful launches (fwd, mul_assign_normalize, inv), for reference
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step_unfused()
    fence()
    unfused = time.perf_counter() - t1

    # per-kernel timing with HIP events on the launch stream (roofline leg), after the timed region
    reps = 50
    fused_ms = plan.time_batch(5, a, rhs=b, reps=reps) / reps
    fwd_ms = plan.time_batch(0, a, reps=reps) / reps
    inv_ms = plan.time_batch(1, a, reps=reps) / reps
    mul_ms = plan.time_batch(2, a, rhs=b, reps=reps) / reps
    alg_bytes = 2 * N * 8 * batch                     # 2*N*sizeof(T) = 16384 B per transform
    # Dominant kernel of the timed region = the fused step kernel: 2 transforms per polynomial per launch.
    # roofline.achieved = per-transform algorithmic bytes x transforms per launch / launch time (the kernel
    # itself moves only 3*N*8 bytes per polynomial -- read lhs, read rhs_ntt, write lhs -- which is what
    # `traffic` reports from the rocprofv3 PMC passes).
    fused_alg = 2 * alg_bytes
    achieved = fused_alg / (fused_ms * 1e-3) / 1e9
    # HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE
    # doubled on gfx950) for exactly this kernel and batch: profiles/r01_v3_rocprofv3_summary.txt
    traffic = (2 * 524748.2 + 524288.0) * 1024 if batch == 65536 else None

    if rank == 0:
        units = world * 2 * batch * args.steps        # forward + inverse transforms, all ranks
        out = {
            "metric": "forward+inverse NTTs/sec (prime64, N=1024, batched) per GPU; % HBM roofline",
            "value": units / elapsed, "unit": "NTT/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "prime64 N=1024 p=4611686018427322369 batch=%d per GPU: fwd + "
                                   "mul_assign_normalize (pre-transformed rhs) + inv, device-resident, fused in one "
                                   "kernel (cntt_prime64_mul_ntt_batch)" % batch,
                       "polynomial_size": N, "batch_per_gpu": batch, "modulus": P62,
                       "sharding": "independent batch shards, no collective"},
            "per_gpu_value": units / elapsed / world,
            "unfused_value": units / unfused, "unfused_ms_per_step": 1e3 * unfused / args.steps,
            "roofline": {"bound": "hbm", "kernel": "mul_kernel_wp<u64, LOGN=10, lazy> (fwd + pointwise + inv fused)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": fused_alg, "avg_launch_ms": fused_ms,
                         "transforms_per_launch": 2 * batch,
                         "moved_bytes_frac": 3 * N * 8 * batch / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "fwd_kernel_ms": fwd_ms, "fwd_frac": alg_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "inv_kernel_ms": inv_ms, "inv_frac": alg_bytes / (inv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "pointwise_kernel_ms": mul_ms,
                         "pointwise_frac": 3 * N * 8 * batch / (mul_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, P62)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "NTT/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
