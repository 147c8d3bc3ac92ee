"""conv shapes through the 8-wave halo kernel (LADDER_DISABLE_HALO16=1) with LADDER_HALO_STAGGER=n; prints us per launch."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child():
    import torch
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    L.load()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for name, (N, H, W, Ci, Co) in dict(conv7=(128, 128, 128, 128, 128), conv6=(128, 64, 64, 256, 128), conv5=(128, 32, 32, 256, 256), conv6b=(128, 64, 64, 128, 256)).items():
        x = torch.randn(N, H, W, Ci, device="cuda")
        w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.03
        y = torch.empty(N, H, W, Co, device="cuda")
        pk = torch.empty(L.query("ladder_filter_pack_split_bytes", 9, Ci, Co, 4), dtype=torch.uint8, device="cuda")
        L.call("ladder_filter_pack_split", w.data_ptr(), pk.data_ptr(), 9, Ci, Co, 0, 4, st)
        rec, yrec = torch.empty(512, device="cuda"), torch.empty(512, device="cuda")
        L.call("ladder_absmax_samples", x.data_ptr(), N, H * W * Ci, rec.data_ptr(), st)
        f = lambda: L.call("ladder_conv3x3_split", x.data_ptr(), rec.data_ptr(), pk.data_ptr(), None, y.data_ptr(), yrec.data_ptr(), N, H, W, Ci, Co, 1, 4, st)
        for _ in range(3): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        out[name] = (round(us, 1), round(2.0 * N * H * W * 9 * Ci * Co / us / 1e6, 1), float(y.double().sum()))
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for halo16, stg in [(1, 0), (0, 0), (0, 4), (0, 8), (0, 12), (0, 16), (0, 24), (0, 32)]:
            env = dict(os.environ)
            if not halo16: env["LADDER_DISABLE_HALO16"] = "1"
            env["LADDER_HALO_STAGGER"] = str(stg)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print("halo16" if halo16 else "8wave ", "stagger", stg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
