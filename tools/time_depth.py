"""Depth sampler at BASELINE configs[1] shape: b=7, v=2, 256x256 rays, 32 buckets, 3 samples;
kernel times from the library's own per-group events, beside the module's (with ReLU+Linear)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from pixelsplat_amd import _lib
from pixelsplat_amd.encoder import DepthPredictorMonocular, sample_depths
dev = torch.device('cuda')
b, v, r, s, spp = 7, 2, 256 * 256, 32, 3
net = DepthPredictorMonocular(128, s, 1, False).to(dev)
feat = torch.randn(b, v, r, 128, device=dev, requires_grad=True)
proj = (torch.randn(b, v, r, 2 * s, device=dev) * 2).requires_grad_(True)
near, far = torch.full((b, v), 0.8, device=dev), torch.full((b, v), 60.0, device=dev)
u = torch.rand(b, v, r, 1, spp, device=dev)
lib = _lib.load()
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(4):
    e0 = ev(); d, o, _ = sample_depths(proj, near, far, 1, u, spp, False, 1.0, 1 / 3); e1 = ev()
    gd, go = torch.ones_like(d), torch.ones_like(o)
    e2 = ev(); torch.autograd.backward([d, o], [gd, go]); e3 = ev()
    e4 = ev(); d, o = net(feat, near, far, False, spp); e5 = ev()
    torch.autograd.backward([d, o], [gd, go]); e6 = ev()
    torch.cuda.synchronize()
    print('sampler fwd %.3f ms bwd %.3f ms | module fwd %.3f ms bwd %.3f ms' % (
        e0.elapsed_time(e1), e2.elapsed_time(e3), e4.elapsed_time(e5), e5.elapsed_time(e6)))
import ctypes as C
ng = lib.ps_profile_group_count()
tot, n = (C.c_double * ng)(), (C.c_int64 * ng)()
lib.ps_profile_enable(1)
for it in range(10):
    d, o, _ = sample_depths(proj, near, far, 1, u, spp, False, 1.0, 1 / 3)
    torch.autograd.backward([d, o], [gd, go])
torch.cuda.synchronize()
lib.ps_profile_enable(0)
lib.ps_profile_collect(tot, n)
for i in range(ng):
    if n[i]:
        print('kernel %s: %.4f ms x %d' % (lib.ps_profile_group_name(i).decode(), tot[i] / n[i], n[i]))
rows = b * v * r
print('algorithmic bytes: fwd %.1f MB, bwd %.1f MB' % (
    rows * (2 * s * 4 + spp * 16) / 1e6, rows * (2 * 2 * s * 4 + spp * 12) / 1e6))
