"""Gaussian adapter at BASELINE configs[1] shape: b=7, v=2, 256x256 rays, 1 surface, 3 samples."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from pixelsplat_amd.encoder import GaussianAdapter, GaussianAdapterCfg
from pixelsplat_amd.synthetic import make_cameras
dev = torch.device('cuda')
b, v, h, w, srf, spp = 7, 2, 256, 256, 1, 3
r = h * w
ctx, _ = make_cameras(b, v, 4, (h, w), torch.Generator().manual_seed(0))
net = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 4)).to(dev)
ext, intr = ctx.extrinsics[:, :, None, None, None].to(dev), ctx.intrinsics[:, :, None, None, None].to(dev)
coords = torch.rand(b, v, r, srf, 1, 2, device=dev, requires_grad=True)
depths = (torch.rand(b, v, r, srf, spp, device=dev) * 5 + 0.5).requires_grad_(True)
op = torch.rand(b, v, r, srf, spp, device=dev, requires_grad=True)
raw = torch.randn(b, v, r, srf, 1, 82, device=dev, requires_grad=True)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(4):
    e0 = ev(); g = net(ext, intr, coords, depths, op, raw, (h, w)); e1 = ev()
    gm, gc, gh = torch.ones_like(g.means), torch.ones_like(g.covariances), torch.ones_like(g.harmonics)
    e2 = ev(); torch.autograd.backward([g.means, g.covariances, g.harmonics], [gm, gc, gh]); e3 = ev()
    torch.cuda.synchronize()
    print('adapter fwd %.3f ms  bwd %.3f ms  (G = %d Gaussians)' % (e0.elapsed_time(e1), e2.elapsed_time(e3), g.means.numel() // 3))
