"""Round 6: hardware counters of the fused forward of a projected pair (up2proj_fused_fwd_kernel) at the conv2d_7 / conv2d_6 shapes of BASELINE configs[2].
   rocprofv3 --kernel-trace --pmc <set> -d out -- python3 profiles/tools/r6_fused_pmc_probe.py ;  ... --show out*/**/*.db"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import r5_pmc_probe as P
P.KERNELS = ("up2proj_fused", "gemm_f32_kernel", "up2proj_fwd_combine")

def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--show":
        return P.show(sys.argv[2:])
    import torch
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    L.load()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    B = 128
    for (H, Cin, Cout) in ((64, 128, 128), (32, 256, 128)):
        N9 = 9 * Cout
        x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
        wcatT = torch.empty(N9, Cin, device="cuda"); y = torch.empty(B, 2 * H, 2 * H, Cout, device="cuda"); bias = torch.randn(Cout, device="cuda")
        L.call("ladder_filter_pack_split", p(w), p(wcatT), 1, N9, Cin, 7, 0, st)
        for _ in range(3):
            L.call("ladder_up2proj_fused_fwd", p(x), p(wcatT), p(bias), p(y), None, None, None, 0, B, H, H, Cin, Cout, 1, None, 0, st)
        torch.cuda.synchronize()
        del x, w, y

if __name__ == "__main__":
    main()
