"""EpipolarSampler mirror (/root/reference/src/model/encoder/epipolar/epipolar_sampler.py:19-166).

Same public surface: attributes num_samples / index_v / transpose_v / transpose_ov
(non-persistent buffers), methods forward / generate_image_rays / transpose / collect, return
type EpipolarSampling with the same eight fields (`features` gathered on first access when the
fused path did not need it).  forward() is one HIP launch for the whole
geometry (ps_epipolar_geometry) plus one for the feature gather (ps_epipolar_gather)."""
from __future__ import annotations

from typing import Callable

import torch
from torch import Tensor, nn

from ..epipolar import EpipolarGeometry, gather_features, sample_geometry


class EpipolarSampling:
    """The reference's record of eight tensors (epipolar_sampler.py:19-29), same names, same constructor
    keywords.  `features` is the one tensor the fused path never needs (0.94 GB at the paper shape): when
    the producer did not materialise it, it is gathered on FIRST ACCESS from the feature map the sampling
    was made from (`lazy_features`) -- a consumer such as the reference's visualiser, which reads
    `sampling.features` out of the `visualization_dump`, gets the tensor either way and the training step
    never pays for it."""
    FIELDS = ("features", "valid", "xy_ray", "xy_sample", "xy_sample_near", "xy_sample_far", "origins",
              "directions")

    def __init__(self, features: Tensor | None, valid: Tensor, xy_ray: Tensor, xy_sample: Tensor,
                 xy_sample_near: Tensor, xy_sample_far: Tensor, origins: Tensor, directions: Tensor,
                 lazy_features: Callable[[], Tensor] | None = None) -> None:
        self._features = features            # [batch, view, other_view, ray, sample, channel] | None
        self._lazy = lazy_features if features is None else None
        self.valid = valid                    # [batch, view, other_view, ray] bool
        self.xy_ray = xy_ray                  # [batch, view, ray, 2]
        self.xy_sample = xy_sample            # [batch, view, other_view, ray, sample, 2]
        self.xy_sample_near = xy_sample_near
        self.xy_sample_far = xy_sample_far
        self.origins = origins                # [batch, view, ray, 3]
        self.directions = directions          # [batch, view, ray, 3]

    @property
    def features(self) -> Tensor | None:
        if self._features is None and self._lazy is not None:
            self._features, self._lazy = self._lazy(), None
        return self._features

    @features.setter
    def features(self, value: Tensor | None) -> None:
        self._features, self._lazy = value, None

    @property
    def features_materialized(self) -> bool:
        return self._features is not None

    def __repr__(self) -> str:
        shapes = {k: (tuple(getattr(self, k).shape) if k != "features" else
                      (tuple(self._features.shape) if self._features is not None else
                       ("lazy" if self._lazy is not None else None))) for k in self.FIELDS}
        return f"EpipolarSampling({shapes})"


def heterogeneous_index(n: int) -> tuple[Tensor, Tensor]:
    """(index_self, index_other): all pairs except self-pairs
    (src/misc/heterogeneous_pairings.py:9-24)."""
    ar = torch.arange(n)
    index_self = ar[:, None].expand(n, n - 1).clone()
    index_other = ar[None, :].expand(n, n).clone() + torch.ones((n, n), dtype=torch.int64).triu()
    return index_self, index_other[:, :-1]


def heterogeneous_index_transpose(n: int) -> tuple[Tensor, Tensor]:
    """Index that "transposes" the heterogeneous pairing (heterogeneous_pairings.py:27-43)."""
    ar = torch.arange(n)
    ones = torch.ones((n, n), dtype=torch.int64)
    index_self = ar[None, :].expand(n, n).clone() + ones.triu()
    index_other = ar[:, None].expand(n, n).clone() - (1 - ones.triu())
    return index_self[:, :-1], index_other[:, :-1]


class EpipolarSampler(nn.Module):
    num_samples: int
    index_v: Tensor
    transpose_v: Tensor
    transpose_ov: Tensor

    def __init__(self, num_views: int, num_samples: int) -> None:
        super().__init__()
        self.num_samples = num_samples
        _, index_v = heterogeneous_index(num_views)
        t_v, t_ov = heterogeneous_index_transpose(num_views)
        self.register_buffer("index_v", index_v, persistent=False)
        self.register_buffer("transpose_v", t_v, persistent=False)
        self.register_buffer("transpose_ov", t_ov, persistent=False)

    # -- HIP path ---------------------------------------------------------------------
    def geometry(self, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                 grid_hw: tuple[int, int]) -> EpipolarGeometry:
        return sample_geometry(extrinsics, intrinsics, near, far, grid_hw, self.num_samples)

    def sampling_from_geometry(self, geo: EpipolarGeometry, grid_hw: tuple[int, int],
                               features: Tensor | None, fmap: Tensor | None = None) -> EpipolarSampling:
        """`features` None + `fmap` (channels-last feature map): the gather is deferred to the first
        read of `.features` (on a detached map: it is a visualisation product, not part of the graph)."""
        h, w = grid_hw
        b, v = geo.origins.shape[:2]
        s = self.num_samples
        dev = geo.origins.device
        ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev),
                                indexing="ij")
        xy = torch.stack(((xs + 0.5) / w, (ys + 0.5) / h), -1).reshape(h * w, 2).float()
        ok = geo.overlaps[..., None]
        xy_min = geo.xy_min.nan_to_num(posinf=0, neginf=0) * ok
        xy_max = geo.xy_max.nan_to_num(posinf=0, neginf=0) * ok
        depth = ((torch.arange(s, device=dev) + 0.5) / s)[:, None]
        half = 0.5 / s
        a, d_ = xy_min[..., None, :], (xy_max - xy_min)[..., None, :]
        lazy = None
        if features is None and fmap is not None:
            src = fmap.detach()
            lazy = lambda: gather_features(src, geo)  # noqa: E731
        return EpipolarSampling(
            features=features, valid=geo.overlaps, xy_ray=xy.expand(b, v, h * w, 2),
            xy_sample=geo.xy_sample, xy_sample_near=a + (depth - half) * d_,
            xy_sample_far=a + (depth + half) * d_, origins=geo.origins,
            directions=geo.directions, lazy_features=lazy)

    def forward(self, images: Tensor, extrinsics: Tensor, intrinsics: Tensor, near: Tensor,
                far: Tensor) -> EpipolarSampling:
        b, v, c, h, w = images.shape
        geo = self.geometry(extrinsics, intrinsics, near, far, (h, w))
        fmap = images.permute(0, 1, 3, 4, 2).contiguous()
        return self.sampling_from_geometry(geo, (h, w), gather_features(fmap, geo))

    # -- index helpers (same semantics as the reference) ---------------------------------
    def generate_image_rays(self, images: Tensor, extrinsics: Tensor, intrinsics: Tensor):
        b, v, _, h, w = images.shape
        geo = sample_geometry(extrinsics, intrinsics, torch.ones((b, v), device=images.device),
                              torch.full((b, v), 2.0, device=images.device), (h, w), 1)
        samp = self.sampling_from_geometry(geo, (h, w), None)
        return samp.xy_ray, geo.origins, geo.directions

    def transpose(self, x: Tensor) -> Tensor:
        b, v, ov, *_ = x.shape
        t_b = torch.arange(b, device=x.device)[:, None, None].expand(b, v, ov)
        t_v = self.transpose_v[None].expand(b, v, ov)
        t_ov = self.transpose_ov[None].expand(b, v, ov)
        return x[t_b, t_v, t_ov]

    def collect(self, target: Tensor) -> Tensor:
        b, v, *_ = target.shape
        index_b = torch.arange(b, device=target.device)[:, None, None].expand(b, v, v - 1)
        index_v = self.index_v[None].expand(b, v, v - 1)
        return target[index_b, index_v]
