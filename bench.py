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

Launch: `python bench.py --gpus N` from a plain shell starts the N ranks itself (fresh child processes, started
before this process touches the GPU; rendezvous on 127.0.0.1); under `python -m torch.distributed.run` the ranks
already exist (WORLD_SIZE is set) and this file is rank code only.
"""
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

P62 = 4611686018427322369   # benches/ntt.rs:115  largest prime = 1 mod 2^16 below 2^62
N = 1024
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "forward+inverse NTTs/sec (prime64, N=1024, batched) per GPU; % HBM roofline"
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc results of the committed build


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


def verify_against_oracle(torch, cntt, plan, n, p, batch, dev, rank, corrupt=False):
    """Correctness gate of the printed number: AFTER the timed region, outside it, the timed step -- the
    fused kernel -- and its three-launch form run once more on a FRESH seeded batch of the same size; polynomials {0, 4097, batch - 1}
    of both results are compared, word for word, with the CPU oracle's fwd / mul_assign_normalize / inv of the same inputs (the call
    pattern of the reference's own test, src/prime64.rs:1254-1266).  The oracle is the checker here, never the thing measured."""
    import numpy as np
    from oracle import pyoracle
    ref = pyoracle.Plan.try_new(n, p, 64)
    a = torch.empty(batch * n, dtype=torch.int64, device=dev)
    b = torch.empty(batch * n, dtype=torch.int64, device=dev)
    cntt.fill_uniform(a, p, 0x5EEDFACE + rank * batch * n)
    cntt.fill_uniform(b, p, 0x5EEDC0DE + rank * batch * n)
    picks = sorted({0, min(4097, batch - 1), batch - 1})
    want = {}
    for i in picks:
        ha = a[i * n:(i + 1) * n].cpu().numpy().view(np.uint64).copy()
        hb = b[i * n:(i + 1) * n].cpu().numpy().view(np.uint64).copy()
        assert int(ha.max()) < p and int(hb.max()) < p
        ref.fwd(ha)
        ref.fwd(hb)
        ref.mul_assign_normalize(ha, hb)
        ref.inv(ha)
        want[i] = ha
    plan.fwd_batch(b)                      # B^, as in the timed loop
    a2 = a.clone()
    plan.mul_ntt_batch(a, b)               # the timed step
    plan.fwd_batch(a2)                     # the same step as three launches
    plan.mul_assign_normalize_batch(a2, b)
    plan.inv_batch(a2)
    torch.cuda.synchronize()
    if corrupt:   # --selftest-corrupt: prove that the gate closes (tests/test_bench_contract.py)
        a[picks[-1] * n + 5] ^= 1
    bad = []
    for i in picks:
        for name, t in (("fused", a), ("unfused", a2)):
            got = t[i * n:(i + 1) * n].cpu().numpy().view(np.uint64)
            if not np.array_equal(got, want[i]):
                bad.append("%s step, polynomial %d: %d of %d words differ" % (name, i, int((got != want[i]).sum()), n))
    same = bool(torch.equal(a, a2))        # and the two device paths agree on the WHOLE batch
    if not same:
        bad.append("fused and three-launch results differ somewhere in the batch")
    del a, b, a2
    return {"verified": not bad, "ran": True, "checker": "oracle/cntt_oracle.c (CPU restatement), polynomials %s of a fresh seeded batch of %d, "
                                           "fused step and fwd / mul_assign_normalize / inv launches; whole batch fused == unfused"
                                           % (picks, batch), "mismatches": bad}


# ---------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no WORLD_SIZE in the environment
# ---------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(world, argv, timeout_s):
    """Start `world` fresh rank processes (this process has not touched the GPU and never will), wait for all of
    them, and return non-zero if any rank failed.  Rank 0 inherits stdout, so its JSON line is this command's."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + timeout_s
    rc = 0
    pending = set(range(world))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % (r, code))
        if rc != 0 or time.monotonic() > deadline:
            if rc == 0:
                rc = 124
                sys.stderr.write("bench.py: ranks did not finish within %d s\n" % timeout_s)
            for r in pending:
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------------------------------------
# rank code
# ---------------------------------------------------------------------------------------------------------
class stdout_to_stderr:
    """gloo announces its connections on the C-level stdout; keep this command's stdout to the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


class Timer:
    """HIP-event timing on the stream the library launches on (torch's current stream of the device: the stream
    `buffer_info` hands to every `_batch` call)."""

    def __init__(self, torch):
        self.torch = torch

    def ramp(self, fn, seconds):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                fn()
            self.torch.cuda.synchronize()

    def ms(self, fn, reps, ramp_s=0.0):
        torch = self.torch
        if ramp_s > 0:
            self.ramp(fn, ramp_s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps


class Sensors:
    """Shader clock and package power of the GPU under test, sampled from a thread while the timed regions run (VERDICT round 3:
    the line must show whether a slow number is a slow box or a slow build).  Source: the amdgpu hwmon files of the card whose PCI
    address torch reports for the device (freq1_input = sclk in Hz, power1_input = package power in microwatts) -- plain file
    reads, no subprocess; falls back to amdsmi's python binding, then to nothing (fields null)."""

    def __init__(self, torch, dev_index, period_s=0.01):
        import threading
        self.period, self.samples, self.phase, self.source = period_s, [], None, None
        self._read = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
            import glob
            for card in glob.glob("/sys/class/drm/card*"):
                d = os.path.join(card, "device")
                if os.path.basename(os.path.realpath(d)) != bdf:
                    continue
                for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                    f, w = os.path.join(h, "freq1_input"), os.path.join(h, "power1_input")
                    if os.path.exists(f) and os.path.exists(w):
                        int(open(f).read()), int(open(w).read())
                        self._read = lambda f=f, w=w: (int(open(f).read()) / 1e6, int(open(w).read()) / 1e6)
                        self.source = "sysfs hwmon %s (freq1_input, power1_input)" % bdf
                if self._read:
                    break
        except Exception:
            self._read = None
        if self._read is None:
            try:
                import amdsmi
                amdsmi.amdsmi_init()
                h = amdsmi.amdsmi_get_processor_handles()[dev_index]

                def rd():
                    pw = amdsmi.amdsmi_get_power_info(h)
                    ck = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                    return float(ck["clk"]), float(pw.get("current_socket_power", pw.get("socket_power")))
                rd()
                self._read, self.source = rd, "amdsmi (current_socket_power, gfx clk)"
            except Exception:
                self._read = None
        self._stop = threading.Event()
        self._thread = None
        if self._read is not None:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def _loop(self):
        while not self._stop.is_set():
            ph = self.phase
            if ph is not None:
                try:
                    mhz, w = self._read()
                    self.samples.append((ph, mhz, w))
                except Exception:
                    pass
            time.sleep(self.period)

    def stop(self):
        self._stop.set()

    def summary(self, phase):
        v = [(m, w) for ph, m, w in self.samples if ph == phase]
        if not v:
            return {"sclk_mhz": None, "power_w": None, "samples": 0}
        return {"sclk_mhz": round(sum(m for m, _ in v) / len(v), 1), "power_w": round(sum(w for _, w in v) / len(v), 1),
                "sclk_mhz_min": min(m for m, _ in v), "power_w_max": max(w for _, w in v), "samples": len(v)}


def pmc_traffic(kernel_key, batch, lib_hash):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by
    tools/profile.sh: FETCH_SIZE doubled for gfx950 + WRITE_SIZE, MI355X_MICROARCH.md section HBM).  The passes run the
    kernel at the bench's own shape (batch 65536); None for any other batch or when no profile is committed.  The file is
    stamped with the source hash of the library it was measured on (`csrc_hash`); the third value returned is True when
    that stamp is missing or differs from the hash compiled into the library that is loaded now (stale profile)."""
    if batch != 65536:
        return None, None, None
    try:
        with open(PMC_FILE) as f:
            doc = json.load(f)
        e = doc[kernel_key]
        stale = not lib_hash or doc.get("csrc_hash") != lib_hash
        return float(e["hbm_bytes_per_launch"]), "profiles/pmc_traffic.json: " + e.get("source", "rocprofv3 --pmc"), stale
    except (OSError, KeyError, ValueError, TypeError):
        return None, None, None


def extra_configs(args, torch, cntt, timer, rank, world, dist, dev, out=None, state=None, sensors=None):
    """The other BASELINE.json configs, timed after the headline region (HIP events, steady-state clocks):
    C3 native64 N=4096, C5 native_binary64 N=2048 (N=1 only), C4 prime64 N=16384 shard-resident (every rank) and --
    with more than one rank -- C4 end to end: scatter from rank 0, transform, gather back."""
    from concrete_ntt_amd import native64, native_binary64, prime64, shard
    out = [] if out is None else out          # filled in place: the watchdog prints what is there if a later leg hangs
    state = {} if state is None else state    # {"phase": name of the running leg, "since": perf_counter when it started}

    def enter(phase):
        state["phase"], state["since"] = phase, time.perf_counter()
        if sensors is not None:
            sensors.phase = phase

    def sensed(phase):
        if sensors is not None:
            sensors.phase = None
            sm = sensors.summary(phase)
            return {"sclk_mhz": sm["sclk_mhz"], "power_w": sm["power_w"]}
        return {}

    def native_case(name, cls, n, batch, binary):
        plan = cls.try_new(n)
        lhs = torch.empty(batch * n, dtype=torch.int64, device=dev)
        rhs = torch.empty(batch * n, dtype=torch.int64, device=dev)
        prod = torch.empty(batch * n, dtype=torch.int64, device=dev)
        cntt.fill_uniform(lhs, 0, 0x5EED0003)
        cntt.fill_uniform(rhs, 0, 0x5EED1003)
        if binary:
            rhs &= 1
        plan.reserve(batch)
        timer.ramp(lambda: plan.negacyclic_polymul_batch(prod, lhs, rhs), 0.5)
        enter(name)
        ms = timer.ms(lambda: plan.negacyclic_polymul_batch(prod, lhs, rhs), 20)
        by = 3 * n * 8 * batch
        e = {"config": name, "workload": "%s negacyclic_polymul N=%d batch=%d, device-resident" % (name, n, batch),
             "ms_per_batch": ms, "value": batch / (ms * 1e-3), "unit": "polymul/s",
             "algorithmic_bytes": by, "roofline_frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        e.update(sensed(name))
        out.append(e)
        del lhs, rhs, prod, plan

    if world == 1 and not args.no_extra:
        for name, cls, n, batch, binary in (("C3 native64::Plan32", native64.Plan32, 4096, 16384, False),
                                            ("C5 native_binary64::Plan32", native_binary64.Plan32, 2048, 65536, True)):
            try:
                native_case(name, cls, n, batch, binary)
            except Exception as e:  # an extra never costs the headline line
                out.append({"config": name, "error": repr(e)})
            torch.cuda.empty_cache()
    if args.no_extra:
        return out

    # C4: prime64 N=16384, 2^20 polynomials over 8 GPUs = 131072 per GPU (16 GiB in place)
    n4, per_gpu = 16384, args.c4_batch
    try:
        plan = prime64.Plan.try_new(n4, P62)
        a = torch.empty(per_gpu * n4, dtype=torch.int64, device=dev)
        cntt.fill_uniform(a, P62, 0x5EED0004 + rank * per_gpu * n4)
        reps = 3
        timer.ramp(lambda: plan.fwd_batch(a), 0.3)
        enter("C4 fwd")
        fwd_ms = timer.ms(lambda: plan.fwd_batch(a), reps)
        s_f = sensed("C4 fwd")
        enter("C4 inv")
        inv_ms = timer.ms(lambda: plan.inv_batch(a), reps)
        s_i = sensed("C4 inv")
        enter("C4 max over ranks")
        t = torch.tensor([fwd_ms, inv_ms], dtype=torch.float64)
        if dist is not None:
            t = t.to(dev) if args.dist_backend == "nccl" else t
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t = t.cpu()
        fwd_ms, inv_ms = float(t[0]), float(t[1])
        by = 2 * n4 * 8 * per_gpu
        out.append({"config": "C4 prime64 N=16384 shard-resident",
                    "workload": "prime64 N=16384 p=%d, %d polynomials per GPU (%.1f GiB in place) x %d GPU(s): fwd and inv "
                                "timed separately, max over ranks" % (P62, per_gpu, per_gpu * n4 * 8 / 2**30, world),
                    "fwd_ms": fwd_ms, "inv_ms": inv_ms,
                    "value": world * 2 * per_gpu / ((fwd_ms + inv_ms) * 1e-3), "unit": "NTT/s", "n_gpus": world,
                    "fwd_roofline_frac": by / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "inv_roofline_frac": by / (inv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "fwd_sclk_mhz": s_f.get("sclk_mhz"), "fwd_power_w": s_f.get("power_w"),
                    "inv_sclk_mhz": s_i.get("sclk_mhz"), "inv_power_w": s_i.get("power_w")})
        if dist is not None:
            # end to end: the whole batch starts and ends on rank 0, which holds the batch AND its own shard twice (the scattered
            # copy and the gather's staging): check the budget first and shrink the leg instead of failing in hipMalloc
            enter("C4 end to end: memory budget")
            del a
            torch.cuda.empty_cache()
            free_b, _ = torch.cuda.mem_get_info(dev)
            ft = torch.tensor([float(free_b)], dtype=torch.float64)
            ft = ft.to(dev) if args.dist_backend == "nccl" else ft
            dist.all_reduce(ft, op=dist.ReduceOp.MIN)
            free_min = float(ft.cpu()[0])
            e2e_per_gpu, shrunk = per_gpu, False
            while e2e_per_gpu > 1 and (world + 3) * e2e_per_gpu * n4 * 8 > 0.9 * free_min:
                e2e_per_gpu //= 2
                shrunk = True
            a = torch.empty(0, dtype=torch.int64, device=dev)
            total = world * e2e_per_gpu
            full = None
            if rank == 0:
                full = torch.empty(total * n4, dtype=torch.int64, device=dev)
                cntt.fill_uniform(full, P62, 0x5EED0C04)
            else:
                full = torch.empty(0, dtype=torch.int64, device=dev)

            def fence():
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()

            for it in range(2):   # first pass creates the point-to-point channels; the second is the one reported
                fence()
                enter("C4 end to end pass %d: scatter" % it)
                t0 = time.perf_counter()
                mine = shard.scatter_batch(full, n4, src=0)
                fence()
                enter("C4 end to end pass %d: fwd" % it)
                t1 = time.perf_counter()
                plan.fwd_batch(mine)
                fence()
                enter("C4 end to end pass %d: gather" % it)
                t2 = time.perf_counter()
                got = shard.gather_batch(mine, n4, total, dst=0, out=full if rank == 0 else None)
                fence()
                t3 = time.perf_counter()
                state.setdefault("done", []).append({"pass": it, "scatter_s": t1 - t0, "compute_s": t2 - t1, "gather_s": t3 - t2})
                del mine, got
            tt = torch.tensor([t1 - t0, t2 - t1, t3 - t2], dtype=torch.float64)
            tt = tt.to(dev) if args.dist_backend == "nccl" else tt
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sc, co, ga = [float(x) for x in tt.cpu()]
            out.append({"config": "C4 prime64 N=16384 end to end",
                        "workload": "%d polynomials (%.0f GiB) on rank 0 -> scatter over %d ranks -> fwd -> gather to rank 0"
                                    % (total, total * n4 * 8 / 2**30, world),
                        "scatter_s": sc, "compute_s": co, "gather_s": ga,
                        "value": total / (sc + co + ga), "unit": "NTT/s (end to end, one fwd per polynomial)",
                        "compute_only_value": total / co, "n_gpus": world,
                        "polynomials_per_gpu": e2e_per_gpu, "shrunk_to_fit_rank0_memory": shrunk})
            del full
        del a
    except Exception as e:
        out.append({"config": "C4 prime64 N=16384", "error": repr(e)})
    torch.cuda.empty_cache()
    enter("done")
    if sensors is not None:
        sensors.phase = None
    return out


def dry_run(args, world, rank):
    """Rehearsal of the launch / rendezvous / max-over-ranks / one-JSON-line plumbing with NO transform executed (for the
    CPU tests: this container has no GPU and there is no CPU compute path).  The line says so and carries no value."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo")
            dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "NTT/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u64", "data": "none", "dry_run": True,
                          "config": {"workload": "DRY RUN: launch plumbing only, no transform executed"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is used\n" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)

    # This command's stdout carries exactly ONE line (rank 0's JSON).  RCCL prints a version banner on the C-level
    # stdout when the first communicator comes up, gloo announces its connections, and under torch.distributed.run
    # every rank shares the launcher's stdout: from here on fd 1 is stderr for the whole process; the line goes to the
    # saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import concrete_ntt_amd as cntt
    from concrete_ntt_amd import prime64

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(ndev, 1)   # rehearsal mode: ranks may share a GPU
    elif local_rank >= ndev:
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d are visible (one rank per GPU with nccl)"
                         % (rank, local_rank, ndev))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            with stdout_to_stderr():
                dist.init_process_group(backend="gloo")
                dist.barrier()
        assert dist.get_world_size() == world

    batch = args.batch
    plan = prime64.Plan.try_new(N, P62)
    dev = torch.device("cuda", local_rank)
    timer = Timer(torch)
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
    sensors = Sensors(torch, local_rank)
    timer.ramp(step, args.ramp_seconds)
    for _ in range(args.warmup):
        step()
    fence()
    sensors.phase = "timed"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    sensors.phase = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same step as three API-faithful launches (fwd, mul_assign_normalize, inv), for reference
    timer.ramp(step_unfused, 0.3)   # its own steady state: three kernels with different power draw alternate
    fence()
    sensors.phase = "unfused"
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step_unfused()
    fence()
    unfused = time.perf_counter() - t1
    sensors.phase = None
    if dist is not None:
        t = torch.tensor([unfused], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        unfused = float(t.item())

    # per-kernel timing with HIP events on the launch stream (roofline leg), after the timed region
    reps = 50

    def kernel_ms(tag, fn):   # ~0.3 s of back-to-back launches first: the sensors see the kernel's own steady state
        timer.ramp(fn, 0.3)
        est = timer.ms(fn, 10)
        n = max(reps, int(60.0 / max(est, 1e-3)) + 1)    # at least 60 ms under the sensors (10 ms sampling period)
        sensors.phase = tag
        ms = timer.ms(fn, n)
        sensors.phase = None
        return ms
    fused_ms = kernel_ms("fused", step)
    fwd_ms = kernel_ms("fwd", lambda: plan.fwd_batch(a))
    inv_ms = kernel_ms("inv", lambda: plan.inv_batch(a))
    mul_ms = kernel_ms("pointwise", lambda: plan.mul_assign_normalize_batch(a, b))
    alg_bytes = 2 * N * 8 * batch                     # 2*N*sizeof(T) = 16384 B per transform
    # Dominant kernel of the timed region = the fused step kernel (2 transforms per polynomial per launch).  It moves
    # 3*N*8 bytes per polynomial -- read lhs, read rhs_ntt, write lhs -- and THAT is what roofline.achieved / frac are
    # computed from (the rocprofv3 PMC passes measure the same figure: `traffic`).  The per-transform accounting of
    # SURVEY 8(d) (2 x 16384 B per polynomial for a kernel that moves 24576 B) is kept as `per_transform_frac`, and the
    # stand-alone fwd / inv kernels -- where moved = algorithmic -- as fwd_frac / inv_frac.
    moved = 3 * N * 8 * batch
    lib_hash = cntt.build_info().get("csrc_hash")
    traffic, traffic_src, traffic_stale = pmc_traffic("mul_kernel_wp<u64, 10, 0, 768, 3>", batch, lib_hash)
    achieved = moved / (fused_ms * 1e-3) / 1e9
    per_transform = 2 * alg_bytes / (fused_ms * 1e-3) / 1e9

    # correctness gate (outside every timed region): rank-local, the line carries the AND over ranks
    try:
        verdict = verify_against_oracle(torch, cntt, plan, N, P62, batch, dev, rank, corrupt=args.selftest_corrupt)
    except Exception as e:
        # the checker itself could not run (no compiler on the box, out of memory for the fresh batch ...): the line says so
        # ("verified": false, "ran": false) but keeps its measurement; only a MISMATCH voids the number
        verdict = {"verified": False, "ran": False, "checker": "oracle/cntt_oracle.c", "mismatches": [],
                   "error": "verification could not run: %r" % (e,)}
    if dist is not None:
        t = torch.tensor([1.0 if verdict["verified"] else 0.0], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if float(t.item()) < 1.0 and verdict["verified"]:
            verdict = dict(verdict, verified=False, mismatches=["another rank's verification failed or could not run"])

    def write_line(extras, with_cpu_baseline):
        if rank != 0:
            return
        units = world * 2 * batch * args.steps        # forward + inverse transforms, all ranks
        out = {
            "metric": METRIC,
            # a number whose results do not match the oracle is not a number: value null, status 4 (see the end of run_rank)
            "value": units / elapsed if not verdict["mismatches"] else None, "unit": "NTT/s",
            "verified": verdict["verified"], "verification": verdict,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "prime64 N=1024 p=4611686018427322369 batch=%d per GPU: fwd + "
                                   "mul_assign_normalize (pre-transformed rhs) + inv, device-resident, fused in one "
                                   "kernel (cntt_prime64_mul_ntt_batch)" % batch,
                       "polynomial_size": N, "batch_per_gpu": batch, "modulus": P62,
                       "sharding": "independent batch shards, no collective",
                       "launcher": "torchrun" if os.environ.get("TORCHELASTIC_RUN_ID") else
                                   ("bench.py --gpus" if world > 1 else "single process"),
                       "dist_backend": args.dist_backend if dist is not None else None,
                       "rccl_ranks": world if (dist is not None and args.dist_backend == "nccl") else 0},
            "per_gpu_value": units / elapsed / world,
            "unfused_value": units / unfused, "unfused_ms_per_step": 1e3 * unfused / args.steps,
            # the three kernels of the unfused step timed alone, each at its own steady-state clock (roofline.sensors): the step
            # alternates them, so its time is their sum at the clock the MIX settles at -- no launch gap (tools/trace_gaps.py)
            "unfused_kernel_sum_ms": fwd_ms + inv_ms + mul_ms,
            "fwd_inv_standalone_value": world * 2 * batch / ((fwd_ms + inv_ms) * 1e-3),   # API-faithful fwd + inv launches, NTT/s
            "roofline": {"bound": "valu/power", "priced_against": "hbm",
                         "kernel": "mul_kernel_wp<u64, LOGN=10, lazy> (fwd + pointwise + inv fused)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_stale": traffic_stale, "csrc_hash": lib_hash,
                         "limiter": "VALU issue of the 62-bit Shoup butterflies under the package power cap (DESIGN.md 5): "
                                    "achieved / peak are HBM bytes actually moved against the 8 TB/s HBM roofline",
                         "moved_bytes_per_launch": moved, "avg_launch_ms": fused_ms,
                         "transforms_per_launch": 2 * batch,
                         "per_transform_bytes_per_launch": 2 * alg_bytes,
                         "per_transform_achieved": per_transform,
                         "per_transform_frac": per_transform / HBM_PEAK_GBS,
                         "fwd_kernel_ms": fwd_ms, "fwd_frac": alg_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "inv_kernel_ms": inv_ms, "inv_frac": alg_bytes / (inv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "pointwise_kernel_ms": mul_ms,
                         "pointwise_frac": 3 * N * 8 * batch / (mul_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # shader clock / package power while the timed region ran (a slow box shows here, not in the build)
                         "sclk_mhz": sensors.summary("timed")["sclk_mhz"], "power_w": sensors.summary("timed")["power_w"],
                         "sensors": {"source": sensors.source, "period_s": sensors.period, "timed_region": sensors.summary("timed"),
                                     "unfused_step": sensors.summary("unfused"),
                                     "fused_kernel": sensors.summary("fused"), "fwd_kernel": sensors.summary("fwd"),
                                     "inv_kernel": sensors.summary("inv"), "pointwise_kernel": sensors.summary("pointwise")}},
            "configs": extras,
            "extras_watchdog": any("watchdog" in str(c.get("error", "")) for c in extras),
        }
        if world == 1 and not args.no_cpu_baseline and with_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, P62)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "NTT/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    # The extras (other BASELINE configs; with several ranks also a 16 GiB-per-rank scatter / gather through RCCL) run AFTER
    # the timed region and must never cost the headline line: a watchdog prints the line without them and ends the rank
    # if they have not finished in time (a collective that never returns cannot be interrupted from Python).
    import threading
    emitted = threading.Lock()

    def emit(extras, with_cpu_baseline=True):
        if not emitted.acquire(blocking=False):
            return
        write_line(extras, with_cpu_baseline)

    extras, xstate = [], {}

    def on_timeout():
        # what finished stays on the line; the entry says which leg was running and for how long (per-phase seconds of an
        # end-to-end pass that completed are under `phases_done`)
        since = xstate.get("since")
        sys.stderr.write("bench.py rank %d: extras still running after %d s (in %r) -- watchdog\n"
                         % (rank, args.extras_timeout, xstate.get("phase")))
        emit(list(extras) + [{"config": "extras", "error": "not finished after %d s (watchdog); the headline numbers above were "
                                                           "complete" % args.extras_timeout,
                              "running_phase": xstate.get("phase"),
                              "seconds_in_phase": (time.perf_counter() - since) if since else None,
                              "phases_done": xstate.get("done", [])}], with_cpu_baseline=False)
        # Status: 0 by default -- the line is complete and says so itself (top-level "extras_watchdog": true and the entry above);
        # a non-zero status would make a launcher (torch.distributed.run, the driver) discard a valid headline because a leg
        # AFTER the timed region hung.  --extras-strict turns the hang into status 3 for callers that want it loud.
        os._exit(3 if args.extras_strict else 0)

    watchdog = threading.Timer(args.extras_timeout, on_timeout)
    watchdog.daemon = True
    watchdog.start()
    try:
        extra_configs(args, torch, cntt, timer, rank, world, dist, dev, out=extras, state=xstate, sensors=sensors)
    except Exception as e:
        extras.append({"config": "extras", "error": repr(e), "running_phase": xstate.get("phase")})
    watchdog.cancel()
    sensors.stop()
    emit(extras)
    if watchdog.is_alive():
        watchdog.join(timeout=1.0)   # a timer that already fired is finishing its write: never close the descriptor under it
    os.close(json_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if verdict["mismatches"]:
        sys.stderr.write("bench.py rank %d: results differ from the oracle: %s\n" % (rank, verdict["mismatches"]))
        sys.exit(4)
    if not verdict["verified"]:
        sys.stderr.write("bench.py rank %d: %s\n" % (rank, verdict.get("error")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 0.12 s timed region: long enough for an SMI sampler to see it
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="polynomials per GPU")
    ap.add_argument("--c4-batch", type=int, default=131072,
                    help="polynomials per GPU of the C4 extra (prime64 N=16384; 131072 = 2^20 / 8 GPUs = 16 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 / C4 / C5 extras (the `configs` array)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one rank per GPU) is what the driver uses; gloo lets several ranks share one "
                         "GPU to rehearse the multi-process path on a 1-GPU box")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even with one rank (a 1-rank RCCL rehearsal of the multi-GPU legs; "
                         "needs RANK / WORLD_SIZE / MASTER_* in the environment, e.g. torch.distributed.run)")
    ap.add_argument("--ramp-seconds", type=float, default=2.0,
                    help="untimed back-to-back steps before the W warm-up steps, so that the timed region "
                         "runs at the steady-state DVFS clock instead of inside the ramp from idle")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous plumbing only (gloo, no GPU, no transform); prints a line with value null")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="seconds the launcher waits for its ranks")
    ap.add_argument("--selftest-corrupt", action="store_true",
                    help="testing only: flip one bit of the verified batch before it is compared with the oracle -- the line must come out "
                         "with value null, verified false and exit status 4")
    ap.add_argument("--extras-strict", action="store_true",
                    help="exit with status 3 (instead of 0) when the extras watchdog fires")
    ap.add_argument("--extras-timeout", type=int, default=420,
                    help="seconds the extras (C3 / C4 / C5 legs after the timed region) may take before a watchdog prints the "
                         "line without them")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above has imported torch or touched HIP.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    run_rank(args)


if __name__ == "__main__":
    main()
