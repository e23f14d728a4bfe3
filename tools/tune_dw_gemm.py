"""Is a tuned library GEMM faster than ps_gemm_tn_f32 for the weight-gradient shapes (k = 57 344)?"""
import os, sys, time
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1",
                  PYTORCH_TUNABLEOP_FILENAME="/tmp/dw.csv", PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="15")
import torch
sys.path.insert(0, '/root/repo')
from pixelsplat_amd.epipolar import gemm_tn
dev = torch.device('cuda')
R = 57344
for m, n in ((592, 128), (128, 592)):
    a = torch.randn(R, m, device=dev); b = torch.randn(R, n, device=dev)
    def t(fn, it=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
    lib = t(lambda: a.T @ b)
    mine = t(lambda: gemm_tn(a, b))
    print(f"dW [{m}x{R}]x[{R}x{n}]: tuned library {lib:.1f} us, ps_gemm_tn_f32 {mine:.1f} us")
print(open('/tmp/dw0.csv').read() if os.path.exists('/tmp/dw0.csv') else 'no file')
