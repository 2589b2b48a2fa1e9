#!/usr/bin/env python3
"""Package power / shader clock trace of the headline kernels under sustained load (rocm-smi sampled every 0.2 s from a
child process while the kernels loop), one phase per kernel:
    fused mul_ntt N=1024 | fwd N=1024 | inv N=1024 | fwd N=16384 | fused N=4096 | tools/ntt_lab (ALU-only ablation) | idle
Output: one line per sample  `t_s  phase  sclk_MHz  power_W`  plus per-phase means.
    python3 tools/power_trace.py [seconds_per_phase] > gpurun_out/power_trace.txt"""
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
from concrete_ntt_amd import prime64  # noqa: E402

P62 = 4611686018427322369
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
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


def loop(name, fn, secs):
    phase[0] = name
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    return (time.perf_counter() - t0) / n


def main():
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    res = {}
    plan = prime64.Plan.try_new(1024, P62)
    a = torch.empty(65536 * 1024, dtype=torch.int64, device="cuda")
    b = torch.empty_like(a)
    cntt.fill_uniform(a, P62, 1)
    cntt.fill_uniform(b, P62, 2)
    plan.fwd_batch(b)
    res["fused_mul_ntt_n1024"] = loop("fused_mul_ntt_n1024", lambda: plan.mul_ntt_batch(a, b), SECS)
    res["fwd_n1024"] = loop("fwd_n1024", lambda: plan.fwd_batch(a), SECS)
    res["inv_n1024"] = loop("inv_n1024", lambda: plan.inv_batch(a), SECS)
    big = prime64.Plan.try_new(16384, P62)
    res["fwd_n16384"] = loop("fwd_n16384", lambda: big.fwd_batch(a), SECS) / 4096 * 65536   # per 65536-polynomial-equivalents of bytes
    p4 = prime64.Plan.try_new(4096, P62)
    res["fused_mul_ntt_n4096"] = loop("fused_mul_ntt_n4096", lambda: p4.mul_ntt_batch(a, b), SECS)
    del a, b
    torch.cuda.empty_cache()
    lab = os.path.join(ROOT, "tools", "ntt_lab")
    if os.path.exists(lab):
        phase[0] = "ntt_lab(ALU-only ablation etc.)"
        lab_out = subprocess.run([lab], capture_output=True, text=True, timeout=200).stdout
    else:
        lab_out = "(tools/ntt_lab not built)"
    phase[0] = "idle"
    time.sleep(1.5)
    stop.set()
    th.join(2)
    print("# t_s phase sclk_MHz package_W   (rocm-smi every 0.2 s; %s)" % cntt.version())
    for t, ph, clk, w in samples:
        print("%7.2f %-34s %5d %7.1f" % (t, ph, clk, w))
    print("# per-phase means (samples after the first second of the phase)")
    for ph in dict.fromkeys(s[1] for s in samples):
        rows = [s for s in samples if s[1] == ph]
        t_first = rows[0][0]
        rows = [s for s in rows if s[0] - t_first >= 1.0] or rows
        print("# %-34s sclk %6.0f MHz  power %7.1f W  (%d samples)%s" % (
            ph, sum(s[2] for s in rows) / len(rows), sum(s[3] for s in rows) / len(rows), len(rows),
            "  %.1f us per launch" % (res[ph] * 1e6) if ph in res else ""))
    print("# tools/ntt_lab output (its own clock stamps per kernel):")
    for ln in lab_out.splitlines():
        print("#   " + ln)


if __name__ == "__main__":
    main()
