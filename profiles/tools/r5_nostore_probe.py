import os, sys
import torch
sys.path.insert(0, "/root/repo")
from ladder_latent_data_distribution_modelling_amd import _lib as L
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
def timeit(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
for (name, M, K, N) in (("c7 F", 524288, 128, 1152), ("c6 F", 131072, 256, 1152), ("c7 B-shape on 32x32 kernel", 524288, 1152, 128)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); z = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * K * N
    t1 = timeit(lambda: L.call("ladder_dense_fwd", p(x), p(w), None, p(z), M, K, N, 0, p(ws), ws.numel(), st))
    t0 = timeit(lambda: L.call("ladder_dense_fwd", p(x), p(w), None, None, M, K, N, 0, p(ws), ws.numel(), st))
    print("%s: with stores %.1f us (%.1f TF), without %.1f us (%.1f TF)" % (name, t1, fl / t1 * 1e-6, t0, fl / t0 * 1e-6))
    del x, w, z
