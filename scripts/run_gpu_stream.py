"""Device STREAM-like check: copy and triad bandwidth of this MI355X with torch ops on 2 GiB arrays."""
import torch, json
n = 1 << 28
a = torch.ones(n, dtype=torch.float64, device='cuda'); b = torch.empty_like(a); c = torch.full_like(a, 2.0)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms_copy = t(lambda: b.copy_(a)); ms_triad = t(lambda: torch.add(a, c, alpha=3.0, out=b))
print(json.dumps({'copy_GBs': 2 * n * 8 / ms_copy / 1e6, 'triad_GBs': 3 * n * 8 / ms_triad / 1e6}))
