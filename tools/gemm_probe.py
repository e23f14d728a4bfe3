"""fp32 GEMM shapes of path (A) on hipBLASLt vs rocBLAS (torch backends), MI355X."""
import os, sys, time, torch
dev = torch.device('cuda')
R = 57344
shapes = [("x@Mq^T   [R,128]x[128,512]", (R, 128), (128, 512), False),
          ("fbar@N^T [R,512]x[512,128]", (R, 512), (512, 128), False),
          ("dW=G^T@x [512,R]x[R,128]", (512, R), (R, 128), False),
          ("dW=F^T@g [128,R]x[R,512]", (128, R), (R, 512), False),
          ("x@Mu^T   [R,128]x[128,80]", (R, 128), (128, 80), False)]
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for backend in ("cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(backend)
    except Exception as e:
        print("backend", backend, "unavailable", e); continue
    for name, sa, sb, _ in shapes:
        a = torch.randn(*sa, device=dev); b = torch.randn(*sb, device=dev)
        us = bench(lambda: a @ b)
        fl = 2.0 * sa[0] * sa[1] * sb[1]
        # transposed-storage variants (what autograd produces: a.T views)
        at = torch.randn(sa[1], sa[0], device=dev).t(); bt = torch.randn(sb[1], sb[0], device=dev).t()
        us_t = bench(lambda: at @ b); us_bt = bench(lambda: a @ bt)
        print(f"{backend:9s} {name:32s} {us:8.1f} us {fl/us/1e6:6.1f} TF | A^T-stored {us_t:8.1f} | B^T-stored {us_bt:8.1f}")
