"""Round 5: what do the ROCm library GEMMs (torch.mm -> hipBLASLt / rocBLAS sgemm, strict fp32) reach at the shapes of the projected decoder pairs?
A yardstick for the in-tree dense kernels (csrc/igemm.hip); the product path does not call the libraries."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False

def timeit(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

for (name, H, Cin, Cout) in (("conv2d_7", 64, 128, 128), ("conv2d_6", 32, 256, 128), ("conv2d_5", 16, 256, 256), ("conv2d_4", 8, 512, 256)):
    M, N9 = 128 * H * H, 9 * Cout
    x = torch.randn(M, Cin, device="cuda"); w = torch.randn(Cin, N9, device="cuda"); z = torch.empty(M, N9, device="cuda")
    wT = w.t().contiguous(); dx = torch.empty(M, Cin, device="cuda"); dw = torch.empty(Cin, N9, device="cuda")
    fl = 2.0 * M * Cin * N9
    t = timeit(lambda: torch.mm(x, w, out=z)); print("%s fwd   [%d,%d]x[%d,%d]  %8.1f us %6.1f TF" % (name, M, Cin, Cin, N9, t, fl / t * 1e-6))
    t = timeit(lambda: torch.mm(z, wT, out=dx)); print("%s bwd   [%d,%d]x[%d,%d]  %8.1f us %6.1f TF" % (name, M, N9, N9, Cin, t, fl / t * 1e-6))
    t = timeit(lambda: torch.mm(x.t(), z, out=dw)); print("%s wgrad [%d,%d]^T x[%d,%d] %8.1f us %6.1f TF" % (name, M, Cin, M, N9, t, fl / t * 1e-6))
    del x, w, z, wT, dx, dw
    torch.cuda.empty_cache()
