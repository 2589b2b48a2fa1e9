#!/usr/bin/env python3
"""Package power / shader clock per kernel family (rocm-smi sampled every 0.2 s while one kernel loops): which kernels sit at the
1400 W cap (clock pulled below 2400 MHz: joules per butterfly is the ceiling) and which run at full clock under the cap
(stalls / HBM are the ceiling: structure still matters).  One line per phase:
    phase  ms_per_launch  frac_of_8TB/s  sclk_MHz  package_W
    python3 tools/power_trace2.py [seconds_per_phase] > gpurun_out/power_trace2.txt"""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
from concrete_ntt_amd import native64, native_binary64, prime32, prime64  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
phase = ["start"]
samples = []
stop = threading.Event()


def sampler():
    t0 = time.perf_counter()
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            out = ""
        mp = re.search(r"Power \(W\):\s*([\d.]+)", out)
        mc = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
        samples.append((time.perf_counter() - t0, phase[0], int(mc.group(1)) if mc else -1, float(mp.group(1)) if mp else -1.0))
        time.sleep(0.2)


results = []


def loop(name, fn, nbytes):
    phase[0] = name
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECS:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = (time.perf_counter() - t0) / n
    results.append((name, dt, nbytes / dt / 8e12))
    print("# done %s" % name, file=sys.stderr, flush=True)


def prime_phases(mod, bits, p, tag, sizes, mib=1024):
    dt = torch.int64 if bits == 64 else torch.int32
    word = bits // 8
    a = torch.empty((mib << 20) // word, dtype=dt, device="cuda")
    cntt.fill_uniform(a, p, 1)
    for n in sizes:
        plan = mod.Plan.try_new(n, p)
        loop("%s_fwd_n%d" % (tag, n), lambda: plan.fwd_batch(a), 2 * a.numel() * word)
        loop("%s_inv_n%d" % (tag, n), lambda: plan.inv_batch(a), 2 * a.numel() * word)
        del plan
    del a
    torch.cuda.empty_cache()


def native_phase(cls, tag, n, batch, binary):
    plan = cls.try_new(n)
    lhs = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    rhs = torch.empty_like(lhs)
    prod = torch.empty_like(lhs)
    cntt.fill_uniform(lhs, 0, 5)
    cntt.fill_uniform(rhs, 0, 6)
    if binary:
        rhs &= 1
    plan.reserve(batch)
    loop("%s_n%d" % (tag, n), lambda: plan.negacyclic_polymul_batch(prod, lhs, rhs), 3 * batch * n * 8)
    del plan, lhs, rhs, prod
    torch.cuda.empty_cache()


def main():
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    prime_phases(prime32, 32, 1073479681, "u32_30bit", [1024, 4096, 16384, 32768])
    prime_phases(prime32, 32, 4293918721, "u32_32bit", [1024, 4096, 16384])
    prime_phases(prime64, 64, 1125899904679937, "u64_fp50", [1024, 4096, 16384, 32768])
    prime_phases(prime64, 64, 4611686018427322369, "u64_lazy62", [1024, 2048, 4096, 8192, 16384, 32768])
    prime_phases(prime64, 64, 18446744069414584321, "u64_solinas", [1024, 16384])
    native_phase(native64.Plan32, "native64_plan32", 4096, 16384, False)
    native_phase(native_binary64.Plan32, "native_binary64_plan32", 2048, 65536, True)
    phase[0] = "idle"
    time.sleep(1.0)
    stop.set()
    th.join(2)
    print("# %s; %g s per phase, rocm-smi every 0.2 s, means over samples after the first second of the phase" % (cntt.version(), SECS))
    print("# %-30s %10s %8s %9s %9s" % ("phase", "ms/launch", "of 8TB/s", "sclk_MHz", "package_W"))
    for name, dt, frac in results:
        rows = [s for s in samples if s[1] == name]
        rows = [s for s in rows if s[0] - rows[0][0] >= 1.0] or rows
        if not rows:
            continue
        print("  %-30s %10.3f %8.3f %9.0f %9.1f" % (name, dt * 1e3, frac, sum(s[2] for s in rows) / len(rows),
                                                   sum(s[3] for s in rows) / len(rows)))


if __name__ == "__main__":
    main()
