import torch
def timeit(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
z = torch.empty(524288, 1152, device="cuda")
t = timeit(lambda: z.fill_(1.0)); print("fill 2.4 GB: %.1f us = %.2f TB/s" % (t, z.numel() * 4 / t * 1e-6))
y = torch.empty_like(z)
t = timeit(lambda: y.copy_(z)); print("copy 2.4 GB: %.1f us = %.2f TB/s (read + write)" % (t, 2 * z.numel() * 4 / t * 1e-6))
t = timeit(lambda: z.sum()); print("sum  2.4 GB: %.1f us = %.2f TB/s (read)" % (t, z.numel() * 4 / t * 1e-6))
