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
