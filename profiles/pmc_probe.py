"""PMC probe for the dominant kernel (conv3x3_halo_kernel) at the BASELINE shape (dec.conv7, batch 128) plus a 1 GiB
device copy used to calibrate FETCH_SIZE / WRITE_SIZE on gfx950.  Run once per counter under rocprofv3 --pmc:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -o pmc -- python3 profiles/pmc_probe.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -o pmc -- python3 profiles/pmc_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ladder_latent_data_distribution_modelling_amd import _lib as L

L.load()
st = torch.cuda.current_stream().cuda_stream
# calibration: y = x copy of 1 GiB (reads 1 GiB, writes 1 GiB)
a = torch.rand(1 << 28, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
# dec.conv7 forward: [128,128,128,128] -> 128 channels, 3x3 SAME, leaky (algorithmic bytes: 1.0737 GB in + 1.0737 GB out + 0.6 MB filters)
N, H, W, Ci, Co = 128, 128, 128, 128, 128
x = torch.randn(N, H, W, Ci, device="cuda")
w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.03
bias = torch.zeros(Co, device="cuda")
y = torch.empty(N, H, W, Co, device="cuda")
assert L.query("ladder_conv2d_fwd_kernel_id", N, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 1) == 256128
for _ in range(3):
    L.call("ladder_conv2d_fwd", x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), N, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 1, None, 0, st)
torch.cuda.synchronize()
