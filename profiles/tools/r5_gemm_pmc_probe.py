"""Round 5: hardware counters of the dense kernels at the projected conv2d_7 shapes (forward Z = x . wcat, backward-data dx = D . wcatT, filter gradient
x^T D), with the ROCm library GEMM (torch.mm, strict fp32) at the same shapes beside them as a yardstick.
   rocprofv3 --kernel-trace --pmc <set> -d out -- python3 profiles/tools/r5_gemm_pmc_probe.py ;  ... --show out*/**/*.db"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import r5_pmc_probe as P
P.KERNELS = ("igemm_fwd_kernel", "igemm_wgrad_kernel", "Cijk", "gemm_f32", "gemm_tn_f32", "gemm_nt16_f32")

def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--show":
        return P.show(sys.argv[2:])
    import torch
    torch.backends.cuda.matmul.allow_tf32 = False
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    L.load()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    ws = lambda n: torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")
    M, K, N9 = 128 * 64 * 64, 128, 1152
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N9, device="cuda") * 0.05; z = torch.empty(M, N9, device="cuda")
    wT = w.t().contiguous(); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(K, N9, device="cuda"); db = torch.empty(N9, device="cuda")
    w1 = ws(L.query("ladder_igemm_fwd_workspace_bytes", M, K, N9)); w2 = ws(L.query("ladder_igemm_fwd_workspace_bytes", M, N9, K))
    w3 = ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, K, N9))
    lib = os.environ.get("PROBE_LIBRARY", "1") == "1"
    for _ in range(3):
        L.call("ladder_dense_fwd", p(x), p(w), None, p(z), M, K, N9, 0, p(w1), w1.numel(), st)
        L.call("ladder_dense_fwd_nt", p(z), p(w), None, p(dx), M, N9, K, 0, st)            # the backward-data GEMM as the engine issues it (16x16x4 kernel)
        L.call("ladder_dense_bwd_weight", p(x), p(z), p(dw), p(db), M, K, N9, p(w3), w3.numel(), st)
        if lib:
            torch.mm(x, w, out=z); torch.mm(z, wT, out=dx)
    torch.cuda.synchronize()

if __name__ == "__main__":
    main()
