"""Round 5: what do the ROCm library GEMMs (torch.mm -> hipBLASLt / rocBLAS sgemm, strict fp32) reach at the shapes of the projected decoder pairs?
A yardstick for the in-tree dense kernels (csrc/igemm.hip); the product path does not call the libraries."""
import os, sys
import torch
torch.backends.cuda.matmul.allow_tf32 = False
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ladder_latent_data_distribution_modelling_amd import _lib as L
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")

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
    t = timeit(lambda: L.call("ladder_dense_fwd", p(x), p(w), None, p(z), M, Cin, N9, 0, p(ws), ws.numel(), st)); print("%s fwd   in-tree kernel, same process    %8.1f us %6.1f TF" % (name, t, fl / t * 1e-6))
    t = timeit(lambda: L.call("ladder_dense_fwd_nt", p(x), p(wT), None, p(z), M, Cin, N9, 0, st)); print("%s fwd   in-tree 16x16x4 NT kernel, same process %8.1f us %6.1f TF" % (name, t, fl / t * 1e-6))
    t = timeit(lambda: torch.mm(z, wT, out=dx)); print("%s bwd   [%d,%d]x[%d,%d]  %8.1f us %6.1f TF" % (name, M, N9, N9, Cin, t, fl / t * 1e-6))
    t = timeit(lambda: L.call("ladder_dense_fwd", p(z), p(wT), None, p(dx), M, N9, Cin, 0, p(ws), ws.numel(), st)); print("%s bwd   in-tree kernel, same process    %8.1f us %6.1f TF" % (name, t, fl / t * 1e-6))
    t = timeit(lambda: L.call("ladder_dense_fwd_nt", p(z), p(w), None, p(dx), M, N9, Cin, 0, st)); print("%s bwd   in-tree 16x16x4 NT kernel, same process %8.1f us %6.1f TF" % (name, t, fl / t * 1e-6))
    t = timeit(lambda: torch.mm(x.t(), z, out=dw)); print("%s wgrad [%d,%d]^T x[%d,%d] %8.1f us %6.1f TF" % (name, M, Cin, M, N9, t, fl / t * 1e-6))
    del x, w, z, wT, dx, dw
    torch.cuda.empty_cache()
