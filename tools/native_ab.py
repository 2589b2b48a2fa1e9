#!/usr/bin/env python3
"""Same-process A/B of the whole-product kernels of the native Plan32 kinds: accumulating CRT (default) against the parked-tile kernels
(testing-only switch native_acc = 0), ns per product on 1 GiB operands, results compared.
    python tools/native_ab.py [kind ...] [--n 8192,16384]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import concrete_ntt_amd as cntt  # noqa: E402
from concrete_ntt_amd import native32, native64, native128, native_binary32, native_binary64, native_binary128  # noqa: E402

KINDS = {"native32": (native32.Plan32, 4), "native64": (native64.Plan32, 8), "native128": (native128.Plan32, 16),
         "native_binary32": (native_binary32.Plan32, 4), "native_binary64": (native_binary64.Plan32, 8),
         "native_binary128": (native_binary128.Plan32, 16),
         "native32_52": (native32.Plan52, 4), "native64_52": (native64.Plan52, 8),
         "native_binary32_52": (native_binary32.Plan52, 4), "native_binary64_52": (native_binary64.Plan52, 8)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kinds", nargs="*", default=["native128", "native_binary128"])
    ap.add_argument("--n", default="4096,8192,16384")
    ap.add_argument("--switch", default="native_acc", help="testing-only switch to toggle (1 = first column, 0 = second): native_acc, plan52_via32")
    ap.add_argument("--tag", default="")   # tools/ab_lib.sh appends --tag new|old
    args = ap.parse_args()
    for kind in args.kinds:
        cls, word = KINDS[kind]
        for n in [int(x) for x in args.n.split(",")]:
            batch = max(8, (1 << 30) // (n * word))
            dt = torch.int32 if word == 4 else torch.int64
            wpp = n * (2 if word == 16 else 1)
            lhs = torch.empty(batch * wpp, dtype=dt, device="cuda")
            rhs = torch.empty_like(lhs)
            cntt.fill_uniform(lhs, 0, 11)
            cntt.fill_uniform(rhs, 0, 22)
            if cls.BINARY:
                rhs &= 1
                if word == 16:
                    rhs.view(-1, 2)[:, 1] = 0
            res = {}
            for acc in (1, 0, 1, 0):
                with cntt.debug_switches(**{args.switch: acc}):
                    plan = cls.try_new(n)
                    plan.reserve(batch)
                    out = torch.empty_like(lhs)
                    for _ in range(3):
                        plan.negacyclic_polymul_batch(out, lhs, rhs)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    reps = 10
                    e0.record()
                    for _ in range(reps):
                        plan.negacyclic_polymul_batch(out, lhs, rhs)
                    e1.record()
                    torch.cuda.synchronize()
                    ns = e0.elapsed_time(e1) / reps * 1e6 / batch
                    res.setdefault(acc, []).append((ns, out.clone()))
            same = torch.equal(res[1][0][1], res[0][0][1])
            by = 3 * n * word
            print(args.tag, "%-18s n=%6d  " % (kind, n) + args.switch + "=1 %8.1f / %8.1f ns (%4.1f %%)   =0 %8.1f / %8.1f ns (%4.1f %%)   identical=%s" % (
                res[1][0][0], res[1][1][0], 100 * by / min(r[0] for r in res[1]) / 8000.0,
                res[0][0][0], res[0][1][0], 100 * by / min(r[0] for r in res[0]) / 8000.0, same), flush=True)
            del lhs, rhs, res
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
