"""Round 5: how fast are the EXISTING dense GEMM kernels at the shapes of the 'project, then upsample' formulation of resize x2 -> 3x3 conv
(nine 1x1 convolutions at LOW resolution + an elementwise combination)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ladder_latent_data_distribution_modelling_amd import _lib as L
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
ws = lambda n: torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")

def timeit(fn, reps=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

for (name, M, Cin, Cout) in (("conv2d_7", 128 * 64 * 64, 128, 128), ("conv2d_6", 128 * 32 * 32, 256, 128), ("conv2d_5", 128 * 16 * 16, 256, 256), ("conv2d_4", 128 * 8 * 8, 512, 256)):
    N9 = 9 * Cout
    x = torch.randn(M, Cin, device="cuda"); w = torch.randn(Cin, N9, device="cuda") * 0.05; z = torch.empty(M, N9, device="cuda")
    fl = 2.0 * M * Cin * N9
    w1 = ws(L.query("ladder_igemm_fwd_workspace_bytes", M, Cin, N9))
    t = timeit(lambda: L.call("ladder_dense_fwd", p(x), p(w), None, p(z), M, Cin, N9, 0, p(w1), w1.numel(), st))
    print("%s proj  Z[M=%d, 9Cout=%d] = x[M, %d] w     %8.1f us %6.1f TF" % (name, M, N9, Cin, t, fl / t * 1e-6))
    wT = torch.randn(N9, Cin, device="cuda") * 0.05; dx = torch.empty(M, Cin, device="cuda")
    w2 = ws(L.query("ladder_igemm_fwd_workspace_bytes", M, N9, Cin))
    t = timeit(lambda: L.call("ladder_dense_fwd", p(z), p(wT), None, p(dx), M, N9, Cin, 0, p(w2), w2.numel(), st))
    print("%s bwd   dx[M, %d] = D[M, %d] wT              %8.1f us %6.1f TF" % (name, Cin, N9, t, fl / t * 1e-6))
    dw = torch.empty(Cin, N9, device="cuda"); db = torch.empty(N9, device="cuda")
    w3 = ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, Cin, N9))
    t = timeit(lambda: L.call("ladder_dense_bwd_weight", p(x), p(z), p(dw), p(db), M, Cin, N9, p(w3), w3.numel(), st))
    print("%s wgrad dW[%d, %d] = x^T D                    %8.1f us %6.1f TF" % (name, Cin, N9, t, fl / t * 1e-6))
    # copy bandwidth reference for the combine pass: read Z, write y (4 Cout per low-res pixel)
    y = torch.empty(M * 4, Cout, device="cuda")
    t = timeit(lambda: (y.copy_(z[:, :4 * Cout].reshape(M * 4, Cout))))
    print("%s HBM reference: copy %d MB                         %8.1f us" % (name, y.numel() * 4 // 2 ** 20, t))
    del x, w, z, wT, dx, dw, y
    torch.cuda.empty_cache()
