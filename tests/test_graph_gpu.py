"""hipGraph capture of the hot path (what `bench.py --launch graph` does): the rasterizer in
fixed-capacity mode and the fused epipolar layers, forward + backward, recorded into graphs and
replayed.  A replay must reproduce the eager results -- bit for bit where the kernels are
deterministic (images, radii, every path-(A) tensor, the slot path of the rasterizer backward);
to fp32 reassociation for the float atomics of Gaussians touching > 4 tiles -- and the overflow
of a fixed-capacity list must be reported after the replay."""
import pytest
import torch

from tests.cases import make_workload

pytestmark = pytest.mark.gpu


def _scene(dev, hw=(64, 64), v=3, seed=5):
    ctx, tgt, g, target = make_workload(1, hw, v_ctx=2, v_tgt=v, seed=seed)
    leaves = [t.to(dev).requires_grad_(True) for t in (g.means, g.covariances, g.harmonics, g.opacities)]
    cams = (tgt.extrinsics.reshape(v, 4, 4).to(dev), tgt.intrinsics.reshape(v, 3, 3).to(dev),
            tgt.near.reshape(v).to(dev), tgt.far.reshape(v).to(dev))
    return leaves, cams, target.reshape(v, 3, *hw).to(dev)


@pytest.mark.parametrize("deterministic", [False, True])
def test_rasterizer_step_replayed_from_a_graph_equals_eager(gpu_device, deterministic):
    """`deterministic=True` (PS_FLAG_DETERMINISTIC): the replayed gradients must equal the eager ones BIT FOR BIT
    -- round 5's first form of the mode cleared its slots with hipMemsetAsync, the only memset node of a captured
    step, and gave garbage gradients when replayed (bench.py's step_check caught it; a kernel clears them now)."""
    from pixelsplat_amd.decoder import render_cuda
    from pixelsplat_amd.loss import mse_loss
    from pixelsplat_amd.raster import captured_overflow_flags

    dev = gpu_device
    hw, v = (64, 64), 3
    leaves, cams, target = _scene(dev, hw, v)
    bg = torch.zeros(v, 3, device=dev)

    def step(cap):
        for t in leaves:
            t.grad = None
        img = render_cuda(*cams, hw, bg, *leaves, views_per_scene=v, list_capacity=cap,
                          deterministic=deterministic)
        mse_loss(img, target, 1.0).backward()
        return img.detach()

    cap = 300000
    img_eager = step(cap).clone()
    grads_eager = [t.grad.clone() for t in leaves]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(cap)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for t in leaves:
        t.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        img_static = step(cap)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    flags = captured_overflow_flags(check=True)
    assert flags and not flags[-1][1] and 0 < flags[-1][0] <= cap
    assert torch.equal(img_static, img_eager)
    for t, ref in zip(leaves, grads_eager):
        if deterministic:
            assert torch.equal(t.grad, ref)
        else:
            torch.testing.assert_close(t.grad, ref, rtol=1e-5, atol=1e-8)

    # a list that does not fit: the replay completes, the flag says so afterwards
    for t in leaves:
        t.grad = None
    with torch.cuda.stream(side):
        try:
            step(2048)
        except RuntimeError:
            pass
    torch.cuda.synchronize()
    for t in leaves:
        t.grad = None
    small = torch.cuda.CUDAGraph()
    with torch.cuda.graph(small):
        step(2048)
    small.replay()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="list_capacity"):
        captured_overflow_flags(check=True)

    # two forwards recorded into ONE graph never share a landing buffer: the fitting
    # call's flag is not overwritten by the overflowing one's
    from pixelsplat_amd.raster import release_captured_flags
    release_captured_flags()
    assert captured_overflow_flags(check=False) == []
    for t in leaves:
        t.grad = None
    both = torch.cuda.CUDAGraph()
    with torch.cuda.graph(both):
        step(cap)
        for t in leaves:
            t.grad = None
        step(2048)
    both.replay()
    torch.cuda.synchronize()
    fl = captured_overflow_flags(check=False)
    assert len(fl) == 2 and fl[0][2] == cap and fl[1][2] == 2048
    assert not fl[0][1] and fl[1][1] and fl[0][0] == flags[-1][0]
    release_captured_flags()


def test_epipolar_layers_replayed_from_a_graph_equal_eager_and_are_deterministic(gpu_device):
    """Two fused cross-attention layers sharing a FeatureGradBatch (two-pass feature-map
    gradient), forward + backward: eager twice -> identical bits (no atomics anywhere on the
    path); graph replay -> the same bits again."""
    from pixelsplat_amd.encoder import (EpipolarTransformer, EpipolarTransformerCfg,
                                        ImageSelfAttentionCfg)
    from pixelsplat_amd.epipolar import FeatureGradBatch
    from pixelsplat_amd.synthetic import make_cameras

    dev = gpu_device
    torch.manual_seed(0)
    b, v, c, h, w = 2, 2, 32, 16, 16
    et = EpipolarTransformer(EpipolarTransformerCfg(
        self_attention=ImageSelfAttentionCfg(patch_size=4, num_octaves=4, num_layers=1, num_heads=2,
                                             d_token=32, d_dot=16, d_mlp=64),
        num_octaves=10, num_layers=2, num_heads=4, num_samples=16, d_dot=16, d_mlp=64, downscale=1),
        c, num_context_views=v).to(dev)
    ctx, _ = make_cameras(b, v, 4, (64, 64), torch.Generator().manual_seed(1))
    cams = [t.to(dev) for t in (ctx.extrinsics, ctx.intrinsics, ctx.near, ctx.far)]
    feat = torch.randn(b, v, h, w, c, device=dev).requires_grad_(True)
    params = [p for n, p in et.named_parameters()
              if n.startswith(("transformer.layers", "depth_encoding")) and "self_attention" not in n]

    def step():
        for t in (feat, *params):
            t.grad = None
        geo = et.epipolar_sampler.geometry(*cams, (h, w))
        x = feat.reshape(-1, 1, c)
        batch = FeatureGradBatch()
        for (attn, _ff), folded in zip(et.transformer.layers, et.fold_layers()):
            x = et.fused_block(attn, x, feat, geo, folded=folded, batch=batch)
        x.square().mean().backward()
        return x.detach()

    x0 = step().clone()
    params = [p for p in params if p.grad is not None]   # (view embeddings etc. are not on this path)
    assert len(params) >= 12
    g0 = [t.grad.clone() for t in (feat, *params)]
    x1 = step().clone()
    assert torch.equal(x0, x1)
    for t, ref in zip((feat, *params), g0):
        assert torch.equal(t.grad, ref)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for t in (feat, *params):
        t.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        xs = step()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(xs, x0)
    for t, ref in zip((feat, *params), g0):
        assert torch.equal(t.grad, ref)
