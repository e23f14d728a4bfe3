"""ORACLE (test infrastructure, NOT product code): imports the REFERENCE's own epipolar
modules from /root/reference, unmodified, on CPU (recipe: SURVEY.md Appendix C).  Only usable
in the build container (the GPU box has no /root/reference): it generates the golden
vectors under tests/golden/ and pins oracle/epipolar_ref.py.

Never imported by the product, bench.py's timed region or the -m gpu tests.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
from types import SimpleNamespace

REF = os.environ.get("PIXELSPLAT_REFERENCE", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shim")
_PKGS = ["src", "src.model", "src.model.encoder", "src.model.encoder.epipolar",
         "src.model.encoder.common",
         "src.model.decoder", "src.model.transformer", "src.model.encodings", "src.geometry",
         "src.misc"]


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "src", "model", "encoder", "epipolar"))


def setup(num_context_views: int = 2):
    """Registers bare parent packages (so the heavy __init__ files that pull torchvision /
    e3nn / wandb never run) and sets the global cfg read at epipolar_transformer.py:46."""
    if not available():
        raise RuntimeError(f"reference not found under {REF}")
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    for name in _PKGS:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, *name.split("."))]
            sys.modules[name] = m
    g = importlib.import_module("src.global_cfg")
    g.set_cfg(SimpleNamespace(
        dataset=SimpleNamespace(view_sampler=SimpleNamespace(num_context_views=num_context_views)),
        seed=0))
    return g


def modules(num_context_views: int = 2) -> SimpleNamespace:
    setup(num_context_views)
    imp = importlib.import_module
    return SimpleNamespace(
        sampler=imp("src.model.encoder.epipolar.epipolar_sampler"),
        transformer=imp("src.model.encoder.epipolar.epipolar_transformer"),
        self_attention=imp("src.model.encoder.epipolar.image_self_attention"),
        conversions=imp("src.model.encoder.epipolar.conversions"),
        lines=imp("src.geometry.epipolar_lines"),
        projection=imp("src.geometry.projection"),
        attention=imp("src.model.transformer.attention"),
        tfm=imp("src.model.transformer.transformer"),
        pe=imp("src.model.encodings.positional_encoding"),
        pairings=imp("src.misc.heterogeneous_pairings"),
    )


def adapter_modules() -> SimpleNamespace:
    """The reference's GaussianAdapter stack.  e3nn (absent, unpinned) is replaced by
    oracle/ref_shim/e3nn, i.e. by oracle/adapter_ref.py's restatement of its two functions."""
    setup(2)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    imp = importlib.import_module
    return SimpleNamespace(
        adapter=imp("src.model.encoder.common.gaussian_adapter"),
        gaussians=imp("src.model.encoder.common.gaussians"),
        sh_rotation=imp("src.misc.sh_rotation"),
        projection=imp("src.geometry.projection"),
    )


def loss_modules() -> SimpleNamespace:
    """The reference's LossMse / LossDepth / compute_psnr.  Their modules import the dataset
    package (for a type annotation) and lpips / skimage (for the other metrics in the same
    file); those get attribute-only stand-ins, the loss code itself is the reference's."""
    setup(2)
    for name in ("src.loss", "src.evaluation", "src.dataset"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, *name.split("."))]
            sys.modules[name] = m
    sys.modules["src.dataset"].DatasetCfg = object
    for name, attrs in (("lpips", ("LPIPS",)), ("skimage", ()),
                        ("skimage.metrics", ("structural_similarity",))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, object)
            sys.modules[name] = m
    imp = importlib.import_module
    return SimpleNamespace(
        mse=imp("src.loss.loss_mse"),
        depth=imp("src.loss.loss_depth"),
        metrics=imp("src.evaluation.metrics"),
        decoder=imp("src.model.decoder.decoder"),
    )


class RasterizerRecorder:
    """Recording stand-in for the absent third-party module `diff_gaussian_rasterization`
    (requirements.txt:17): the reference's UNMODIFIED cuda_splatting.py imports it
    (cuda_splatting.py:5-8) and calls it per view (:99-124).  Every call is recorded exactly
    as the reference hands it over (settings + arguments); the image it returns is rendered
    by oracle/raster_ref.c, so what comes out of the reference functions is
    "reference host glue + oracle rasterizer"."""

    def __init__(self, render: bool = True, differentiable: bool = False, keep_calls: bool = True):
        """`differentiable`: the stand-in is an autograd function -- forward = oracle/raster_ref.c,
        backward = its hand-written backward -- so that the reference's decoder, encoder head and
        epipolar transformer can be chained into ONE training step on the CPU with gradients flowing
        end to end (tests/golden/make_connected_golden.py)."""
        self.calls: list[dict] = []
        self.render = render
        self.differentiable = differentiable
        self.keep_calls = keep_calls

    def install(self):
        import typing

        import numpy as np
        import torch
        from torch import nn

        rec = self

        class GaussianRasterizationSettings(typing.NamedTuple):
            image_height: int
            image_width: int
            tanfovx: float
            tanfovy: float
            bg: torch.Tensor
            scale_modifier: float
            viewmatrix: torch.Tensor
            projmatrix: torch.Tensor
            sh_degree: int
            campos: torch.Tensor
            prefiltered: bool
            debug: bool

        class GaussianRasterizer(nn.Module):
            def __init__(self, raster_settings):
                super().__init__()
                self.raster_settings = raster_settings

            def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None,
                        scales=None, rotations=None, cov3D_precomp=None):
                s = self.raster_settings
                npf = lambda t: None if t is None else np.ascontiguousarray(
                    t.detach().cpu().numpy().astype(np.float32))
                call = dict(
                    image_height=int(s.image_height), image_width=int(s.image_width),
                    tanfovx=float(s.tanfovx), tanfovy=float(s.tanfovy), bg=npf(s.bg),
                    scale_modifier=float(s.scale_modifier), viewmatrix=npf(s.viewmatrix),
                    projmatrix=npf(s.projmatrix), sh_degree=int(s.sh_degree),
                    campos=npf(s.campos), campos_stride=tuple(s.campos.stride()),
                    viewmatrix_contiguous=bool(s.viewmatrix.is_contiguous()),
                    means3D=npf(means3D), means2D_shape=tuple(means2D.shape),
                    means2D_requires_grad=bool(means2D.requires_grad), opacities=npf(opacities),
                    shs=npf(shs), colors_precomp=npf(colors_precomp), cov3D_precomp=npf(cov3D_precomp),
                    scales=scales, rotations=rotations)
                h, w = call["image_height"], call["image_width"]
                if rec.render and rec.differentiable and shs is not None and cov3D_precomp is not None:
                    from oracle import raster_ref as R

                    class _Oracle(torch.autograd.Function):
                        @staticmethod
                        def forward(ctx, m3, m2, op, sh_, cov_):
                            st = R.forward(
                                means=call["means3D"], cov6=call["cov3D_precomp"],
                                opacity=call["opacities"][:, 0], view=call["viewmatrix"].reshape(16),
                                proj=call["projmatrix"].reshape(16), campos=call["campos"], bg=call["bg"],
                                H=h, W=w, tanfovx=call["tanfovx"], tanfovy=call["tanfovy"],
                                sh=call["shs"], colors=None, sh_degree=call["sh_degree"])
                            ctx.st = st
                            call["image"] = st.image.copy()
                            call["radii"] = st.radii.copy()
                            call["ambiguous"] = R.ambiguity_mask(st)
                            radii_ = torch.from_numpy(st.radii.copy())
                            ctx.mark_non_differentiable(radii_)
                            return torch.from_numpy(st.image.copy()), radii_

                        @staticmethod
                        def backward(ctx, d_img, _d_radii):
                            g = R.backward(ctx.st, d_img.detach().cpu().numpy().astype(np.float32))
                            f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
                            return (f(g["means3D"]), f(g["means2D"]), f(g["opacity"])[:, None], f(g["sh"]),
                                    f(g["cov6"]))

                    image, radii = _Oracle.apply(means3D, means2D, opacities, shs, cov3D_precomp)
                    if rec.keep_calls:
                        rec.calls.append(call)
                    return image, radii
                if rec.render:
                    from oracle import raster_ref as R

                    st = R.forward(
                        means=call["means3D"], cov6=call["cov3D_precomp"],
                        opacity=call["opacities"][:, 0], view=call["viewmatrix"].reshape(16),
                        proj=call["projmatrix"].reshape(16), campos=call["campos"], bg=call["bg"],
                        H=h, W=w, tanfovx=call["tanfovx"], tanfovy=call["tanfovy"],
                        sh=call["shs"], colors=call["colors_precomp"], sh_degree=call["sh_degree"])
                    call["image"] = st.image.copy()
                    call["radii"] = st.radii.copy()
                    call["ambiguous"] = R.ambiguity_mask(st)
                    image = torch.from_numpy(st.image.copy())
                    radii = torch.from_numpy(st.radii.copy())
                else:
                    image = torch.zeros((3, h, w))
                    radii = torch.zeros(means3D.shape[0], dtype=torch.int32)
                rec.calls.append(call)
                return image, radii

        mod = types.ModuleType("diff_gaussian_rasterization")
        mod.GaussianRasterizationSettings = GaussianRasterizationSettings
        mod.GaussianRasterizer = GaussianRasterizer
        sys.modules["diff_gaussian_rasterization"] = mod
        return self


def decoder_modules(recorder: RasterizerRecorder) -> SimpleNamespace:
    """The reference's decoder host code (cuda_splatting.py, decoder_splatting_cuda.py), the
    spin trajectory of its one rasterizer fixture (scripts/test_splatter.py) and rotate_sh,
    imported unmodified on top of `recorder` (which must stand in for the rasterizer BEFORE
    cuda_splatting.py is imported)."""
    setup(2)
    recorder.install()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    for name in ("src.dataset", "src.visualization", "src.visualization.camera_trajectory"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, *name.split("."))]
            sys.modules[name] = m
    sys.modules["src.dataset"].DatasetCfg = object
    for stale in ("src.model.decoder.cuda_splatting", "src.model.decoder.decoder_splatting_cuda"):
        sys.modules.pop(stale, None)   # re-bind to THIS recorder
    imp = importlib.import_module
    return SimpleNamespace(
        splatting=imp("src.model.decoder.cuda_splatting"),
        decoder_cuda=imp("src.model.decoder.decoder_splatting_cuda"),
        decoder=imp("src.model.decoder.decoder"),
        types=imp("src.model.types"),
        projection=imp("src.geometry.projection"),
        spin=imp("src.visualization.camera_trajectory.spin"),
        sh_rotation=imp("src.misc.sh_rotation"),
    )
