"""EpipolarTransformer mirror
(/root/reference/src/model/encoder/epipolar/epipolar_transformer.py:20-183,
 image_self_attention.py:13-79, src/model/encodings/positional_encoding.py:8-36).

Same constructor `(cfg, d_in)`, same parameter names (depth_encoding.1, transformer.layers.*,
downscaler, upscaler, upscale_refinement.*, view_embeddings) and the same return value
`(features, sampling)`.  The hot part -- sampler geometry, depth, depth encoding, feature
gather and the two cross-attention layers -- runs on the HIP kernels; kv
([b,v,ov,r,s,c], 0.94 GB at the paper config) is never materialised.  When a visualiser
hooks `layer[0].fn.attend` (encoder_visualizer_epipolar.py:52-56) the layer falls back to
the unfused formulation on a materialised kv so the hook sees the softmax output.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import partial
from typing import Optional

import torch
from torch import Tensor, nn

from ..epipolar import (FeatureGradBatch, fold_attention_weights, fused_cross_attention,
                        gather_features, layer_norm, layer_norm_fork)
from .epipolar_sampler import EpipolarSampler, EpipolarSampling
from .transformer import Transformer


@dataclass
class ImageSelfAttentionCfg:
    patch_size: int
    num_octaves: int
    num_layers: int
    num_heads: int
    d_token: int
    d_dot: int
    d_mlp: int


@dataclass
class EpipolarTransformerCfg:
    self_attention: ImageSelfAttentionCfg
    num_octaves: int
    num_layers: int
    num_heads: int
    num_samples: int
    d_dot: int
    d_mlp: int
    downscale: int


class PositionalEncoding(nn.Module):
    """sin(x * 2 pi 2^k + {0, pi/2}) for values in [0, 1] (positional_encoding.py:8-36)."""

    def __init__(self, num_octaves: int):
        super().__init__()
        octaves = torch.arange(num_octaves).float()
        frequencies = (2 * torch.pi * 2 ** octaves)[:, None].expand(num_octaves, 2).contiguous()
        self.register_buffer("frequencies", frequencies, persistent=False)
        phases = torch.tensor([0, 0.5 * torch.pi], dtype=torch.float32)
        self.register_buffer("phases", phases[None].expand(num_octaves, 2).contiguous(),
                             persistent=False)

    def forward(self, samples: Tensor) -> Tensor:
        x = samples[..., None, None] * self.frequencies
        return torch.sin(x + self.phases).flatten(-3)

    def d_out(self, dimensionality: int) -> int:
        return self.frequencies.numel() * dimensionality


class ImageSelfAttention(nn.Module):
    def __init__(self, cfg: ImageSelfAttentionCfg, d_in: int, d_out: int):
        super().__init__()
        self.positional_encoding = nn.Sequential(
            (pe := PositionalEncoding(cfg.num_octaves)), nn.Linear(pe.d_out(2), cfg.d_token))
        self.patch_embedder = nn.Sequential(
            nn.Conv2d(d_in, cfg.d_token, cfg.patch_size, cfg.patch_size), nn.ReLU())
        self.transformer = Transformer(cfg.d_token, cfg.num_layers, cfg.num_heads, cfg.d_dot,
                                       cfg.d_mlp)
        self.resampler = nn.ConvTranspose2d(cfg.d_token, d_out, cfg.patch_size, cfg.patch_size)

    def forward(self, image: Tensor) -> Tensor:
        tokens = self.patch_embedder(image)
        _, _, nh, nw = tokens.shape
        ys, xs = torch.meshgrid(torch.arange(nh, device=image.device),
                                torch.arange(nw, device=image.device), indexing="ij")
        xy = torch.stack(((xs + 0.5) / nw, (ys + 0.5) / nh), -1).float()
        tokens = tokens + self.positional_encoding(xy).permute(2, 0, 1)
        b, c = tokens.shape[:2]
        tokens = self.transformer(tokens.flatten(2).transpose(1, 2))
        tokens = tokens.transpose(1, 2).reshape(b, c, nh, nw)
        return self.resampler(tokens)


class ImageSelfAttentionWrapper(nn.Module):
    def __init__(self, self_attention_cfg: ImageSelfAttentionCfg, d_in: int, d_hidden: int,
                 dropout: float):
        super().__init__()
        self.self_attention = ImageSelfAttention(self_attention_cfg, d_in, d_in)

    def forward(self, x: Tensor, b: int, v: int, h: int, w: int) -> Tensor:
        c = x.shape[-1]
        img = x.reshape(b * v, h, w, c).permute(0, 3, 1, 2)
        img = self.self_attention(img) + img
        return img.permute(0, 2, 3, 1).reshape(b * v * h * w, 1, c)


def _num_context_views_from_reference_cfg() -> Optional[int]:
    try:  # running inside the reference: src/global_cfg.py (epipolar_transformer.py:46)
        from src.global_cfg import get_cfg  # type: ignore

        return int(get_cfg().dataset.view_sampler.num_context_views)
    except Exception:
        return None


class EpipolarTransformer(nn.Module):
    cfg: EpipolarTransformerCfg

    def __init__(self, cfg: EpipolarTransformerCfg, d_in: int,
                 num_context_views: Optional[int] = None) -> None:
        super().__init__()
        if num_context_views is None:
            num_context_views = _num_context_views_from_reference_cfg()
        if num_context_views is None:
            raise ValueError("num_context_views not given and the reference global cfg is absent")
        self.cfg = cfg
        self._side_stream = None
        self.epipolar_sampler = EpipolarSampler(num_context_views, cfg.num_samples)
        if cfg.num_octaves > 0:
            self.depth_encoding = nn.Sequential(
                (pe := PositionalEncoding(cfg.num_octaves)), nn.Linear(pe.d_out(1), d_in))
        feed_forward_layer = partial(ImageSelfAttentionWrapper, cfg.self_attention)
        self.transformer = Transformer(d_in, cfg.num_layers, cfg.num_heads, cfg.d_dot, cfg.d_mlp,
                                       selfatt=False, kv_dim=d_in,
                                       feed_forward_layer=feed_forward_layer)
        self.downscaler = self.upscaler = self.upscale_refinement = None
        if cfg.downscale:
            self.downscaler = nn.Conv2d(d_in, d_in, cfg.downscale, cfg.downscale)
            self.upscaler = nn.ConvTranspose2d(d_in, d_in, cfg.downscale, cfg.downscale)
            self.upscale_refinement = nn.Sequential(
                nn.Conv2d(d_in, d_in * 2, 7, 1, 3), nn.GELU(), nn.Conv2d(d_in * 2, d_in, 7, 1, 3))
        if num_context_views > 2:
            self.view_embeddings = nn.Embedding(num_context_views, d_in)

    @property
    def _kernel_octaves(self) -> int:
        """Octaves the fused kernels run with: num_octaves = 0 (kv = sampled features only,
        epipolar_transformer.py:100-121, re10k_ablation_no_depth_encoding.yaml:29) is one octave
        with zero encoding weights -- the kernel adds exactly 0.0 to every token."""
        return max(self.cfg.num_octaves, 1)

    def _layer_weights(self, attn: nn.Module, view_emb=None) -> dict:
        a = attn.fn
        c = a.to_kv.weight.shape[1]
        if self.cfg.num_octaves > 0:
            depth_w, depth_b = self.depth_encoding[1].weight, self.depth_encoding[1].bias
        else:
            z = a.to_kv.weight.new_zeros((c, 3))
            depth_w, depth_b = z[:, :2], z[:, 2]
        to_out = a.to_out[0] if isinstance(a.to_out, nn.Sequential) else None
        return dict(w_q=a.to_q.weight, w_kv=a.to_kv.weight,
                    w_out=(to_out.weight if to_out is not None else
                           torch.eye(c, device=depth_w.device, dtype=depth_w.dtype)),
                    b_out=(to_out.bias if to_out is not None else None), heads=a.heads,
                    depth_w=depth_w, depth_b=depth_b, view_emb=view_emb)

    def fold_layers(self, view_emb=None) -> list:
        """Folded weight matrices of every cross-attention layer, computed on a side stream:
        they depend on the parameters only, and their ~40 tiny kernels per layer (forward, and
        again in backward -- autograd replays a node on the stream of its forward) then overlap
        the big kernels instead of queueing between them."""
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        side = self._side_stream
        # The parameters enter the side stream's graph through an alias made HERE, on the main stream: autograd
        # replays a node on its forward's stream, so without it the fold's backward hands the parameter gradients
        # to their AccumulateGrad nodes from the side stream -- and under DistributedDataParallel, which creates
        # those nodes at construction (on the stream current then) and keeps them, torch warns "AccumulateGrad
        # node's stream does not match" and synchronises every step (tests/test_ddp_gpu.py).  An alias is
        # metadata only, forward and backward: no kernel.
        def on_main(wd: dict) -> dict:
            return {k: (t.view_as(t) if isinstance(t, Tensor) and t.is_leaf and t.requires_grad else t)
                    for k, t in wd.items()}

        weights = [None if len(attn.fn.attend._forward_hooks) > 0 else on_main(self._layer_weights(attn, view_emb))
                   for attn, _ in self.transformer.layers]
        side.wait_stream(main)
        folds = []
        with torch.cuda.stream(side):
            for wd in weights:
                if wd is None:
                    folds.append(None)        # hooked layers take the unfused path: nothing to fold
                    continue
                f = fold_attention_weights(**wd)
                done = torch.cuda.Event()
                done.record(side)
                for t in f:
                    t.record_stream(main)
                folds.append((*f, done, side))   # fused_layer waits for `done` where it needs f
        return folds

    def fused_layer(self, attn: nn.Module, x: Tensor, fmap: Tensor, geo, view_emb=None,
                    folded=None, batch=None) -> Tensor:
        """PreNorm(Attention)(x, z=kv) (pre_norm.py:34-35, attention.py:54-70) on the HIP path;
        `attn` is one `layer[0]` of `self.transformer.layers`, fmap is channels-last."""
        side = None
        if folded is not None and len(folded) >= 4:
            torch.cuda.current_stream().wait_event(folded[3])
            side = folded[4] if len(folded) > 4 else None
            folded = folded[:3]
        return fused_cross_attention(layer_norm(x, attn.norm), fmap, geo, octaves=self._kernel_octaves,
                                     folded=folded, batch=batch, wgrad_stream=side,
                                     **self._layer_weights(attn, view_emb))

    def fused_block(self, attn: nn.Module, x: Tensor, fmap: Tensor, geo, view_emb=None,
                    folded=None, batch=None) -> Tensor:
        """`PreNorm(Attention)(x, z=kv) + x` (transformer.py:68): fused_layer plus the residual,
        with the residual's gradient folded into the LayerNorm backward."""
        side = None
        if folded is not None and len(folded) >= 4:
            torch.cuda.current_stream().wait_event(folded[3])
            side = folded[4] if len(folded) > 4 else None
            folded = folded[:3]
        xn, xr = layer_norm_fork(x, attn.norm)
        return fused_cross_attention(xn, fmap, geo, octaves=self._kernel_octaves, folded=folded,
                                     batch=batch, wgrad_stream=side,
                                     **self._layer_weights(attn, view_emb)) + xr

    def forward(self, features: Tensor, extrinsics: Tensor, intrinsics: Tensor, near: Tensor,
                far: Tensor, materialize_sampling: bool = False,
                view_shuffle: Optional[Tensor] = None) -> tuple[Tensor, EpipolarSampling]:
        b, v, c, _, _ = features.shape
        if self.downscaler is not None:
            features = self.downscaler(features.flatten(0, 1)).unflatten(0, (b, v))
        _, _, _, h, w = features.shape
        sampler = self.epipolar_sampler
        geo = sampler.geometry(extrinsics, intrinsics, near, far, (h, w))
        fmap = features.permute(0, 1, 3, 4, 2).contiguous()          # channels-last

        view_emb = None
        if v > 2:   # randomly permuted per-view embeddings (epipolar_transformer.py:126-131)
            if view_shuffle is None:
                view_shuffle = torch.randperm(v - 1, device=features.device)
            view_emb = self.view_embeddings(view_shuffle)

        hooked = any(len(layer[0].fn.attend._forward_hooks) > 0 for layer in self.transformer.layers)
        need_kv = hooked or materialize_sampling
        sampled = gather_features(fmap, geo) if need_kv else None
        sampling = sampler.sampling_from_geometry(geo, (h, w), sampled, fmap)   # .features: lazy if None

        x = fmap.reshape(b * v * h * w, 1, c)
        kv = None
        folds = self.fold_layers(view_emb) if features.is_cuda else [None] * len(self.transformer.layers)
        grad_batch = FeatureGradBatch()     # the layers' feature-map gradients share one scatter
        # the map as the attention layers see it: its gradient is produced on a side stream and only
        # waited for where autograd hands it on; the query tokens `x` come from the original map
        fmap_kv = grad_batch.attach(fmap)
        for (attn, ff), folded in zip(self.transformer.layers, folds):
            a = attn.fn
            if len(a.attend._forward_hooks) > 0:
                if kv is None:   # unfused formulation so the hook sees the attention weights
                    kv = sampled
                    if self.cfg.num_octaves > 0:
                        kv = kv + self.depth_encoding(geo.rel_disparity[..., None])
                    if view_emb is not None:
                        kv = kv + view_emb[None, None, :, None, None, :]
                    kv = kv.permute(0, 1, 3, 4, 2, 5).reshape(b * v * h * w, -1, c)
                x = attn(x, z=kv) + x
            else:
                x = self.fused_block(attn, x, fmap_kv, geo, view_emb, folded, grad_batch)
            x = ff(x, b=b, v=v, h=h, w=w) + x
        features = x.reshape(b, v, h, w, c).permute(0, 1, 4, 2, 3)

        if self.upscaler is not None:
            f = self.upscaler(features.flatten(0, 1))
            f = self.upscale_refinement(f) + f
            features = f.unflatten(0, (b, v))
        return features, sampling
