"""A/B of the split-k weight-gradient GEMM (csrc/gemm_tn.hip): PS_GEMM_TN_VARIANT = 0 / 1 / 2,
one process per variant (the switch is read once).  Prints us per call and the error against
a float64 product."""
import os
import sys

import torch

from pixelsplat_amd.epipolar import gemm_tn

dev = torch.device("cuda:0")
shapes = [(57344, 576, 128), (57344, 128, 576), (49152, 592, 128), (65536, 576, 128), (4096, 576, 128)]
for k, m, n in shapes:
    torch.manual_seed(0)
    a = torch.randn(k, m, device=dev)
    b = torch.randn(k, n, device=dev)
    c = gemm_tn(a, b)
    ref = (a.double().T @ b.double())
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    for _ in range(5):
        gemm_tn(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gemm_tn(a, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"variant {os.environ.get('PS_GEMM_TN_VARIANT', 'default')}: k={k} m={m} n={n}: {us:7.1f} us "
          f"({2 * k * m * n / us / 1e6:6.1f} TFLOP/s) rel err {err:.1e}")
    sys.stdout.flush()
