import sys, time, torch
sys.path.insert(0, '/root/repo')
import concrete_ntt_amd as cntt
from concrete_ntt_amd import prime64
P62 = 4611686018427322369
n, batch = 1024, 65536
plan = prime64.Plan.try_new(n, P62)
a = torch.empty(batch * n, dtype=torch.int64, device='cuda'); b = torch.empty_like(a); c = torch.empty_like(a)
cntt.fill_uniform(a, P62, 1); cntt.fill_uniform(b, P62, 2)
def timed(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
nb = a.numel() * 8
for name, fn, streams in (("mul_assign_normalize", lambda: plan.mul_assign_normalize_batch(a, b), 3),
                          ("normalize", lambda: plan.normalize_batch(a), 2),
                          ("torch a.add_(b)", lambda: a.add_(b), 3),
                          ("torch c.copy_(a)", lambda: c.copy_(a), 2),
                          ("torch a.add_(1)", lambda: a.add_(1), 2)):
    ms = timed(fn)
    print("%-24s %.3f ms  %.2f TB/s" % (name, ms, streams * nb / ms / 1e9))
