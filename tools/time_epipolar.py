import sys, time, torch
sys.path.insert(0, '/root/repo')
from pixelsplat_amd.synthetic import make_cameras
from pixelsplat_amd.epipolar import sample_geometry, fused_cross_attention
dev = torch.device('cuda')
b, v, c, h, w, s, heads, dh = 7, 2, 128, 64, 64, 32, 4, 128
gen = torch.Generator().manual_seed(0)
ctx, _ = make_cameras(b, v, 4, (256, 256), gen)
ext, intr, near, far = (t.to(dev) for t in (ctx.extrinsics, ctx.intrinsics, ctx.near, ctx.far))
feat = torch.randn(b, v, h, w, c, device=dev, requires_grad=True)
inner = heads * dh
P = {k: t.to(dev).requires_grad_(True) for k, t in dict(w_q=torch.randn(inner, c) * 0.05, w_kv=torch.randn(2 * inner, c) * 0.05,
     w_out=torch.randn(c, inner) * 0.05, b_out=torch.zeros(c), depth_w=torch.randn(c, 20) * 0.05, depth_b=torch.zeros(c)).items()}
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(3):
    e0 = ev(); geo = sample_geometry(ext, intr, near, far, (h, w), s); e1 = ev()
    x = feat.reshape(-1, 1, c)
    y = fused_cross_attention(x, feat, geo, heads=heads, octaves=10, **P); e2 = ev()
    y.sum().backward(); e3 = ev()
    torch.cuda.synchronize()
    print('geometry %.3f ms  attn fwd %.3f ms  attn bwd %.3f ms' % (e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)))
