"""Zero-edit adoption: swaps the HIP implementations into the REFERENCE's own registries and
module namespaces at import time, so `src/main.py` runs unmodified on MI355X.

    PYTHONPATH=/path/to/this/repo python -m pixelsplat_amd.patch_reference src.main +experiment=re10k

or, from Python, `import pixelsplat_amd.patch_reference as pr; pr.apply()` before the model is
built.  `PIXELSPLAT_HIP=0` leaves the reference untouched (A/B runs with one switch, no new
required config key -- SURVEY.md 5, "config" row).

What is swapped (all by name; the reference resolves every one of them at construction time):

  module `diff_gaussian_rasterization`                      cuda_splatting.py:5-8
  src.model.decoder.DECODERS["splatting_cuda"]              decoder/__init__.py:5-13
  src.model.encoder.encoder_epipolar.EpipolarTransformer    encoder_epipolar.py:15-17, :66-69
  src.model.encoder.encoder_epipolar.DepthPredictorMonocular  :72-77
  src.model.encoder.encoder_epipolar.GaussianAdapter        :78
  src.loss.LOSSES[LossMseCfgWrapper / LossDepthCfgWrapper]  loss/__init__.py:6-10

The replacements keep the reference's constructor signatures, parameter and buffer names (so
`state_dict`s and released checkpoints load unchanged) and return types; see the modules'
docstrings.  Every target is attempted independently and reported, so a partial environment
(e.g. the build container, where the backbones' torchvision is absent) still patches what it
can import.
"""
from __future__ import annotations

import importlib
import os
import sys

_APPLIED: dict[str, str] = {}


def enabled() -> bool:
    return os.environ.get("PIXELSPLAT_HIP", "1").strip().lower() not in ("0", "false", "off", "no")


def _try(report: dict, name: str, fn) -> None:
    try:
        fn()
        report[name] = "patched"
    except Exception as err:   # ImportError of a reference dependency, missing attribute, ...
        report[name] = f"skipped ({type(err).__name__}: {err})"


def apply(force: bool | None = None) -> dict[str, str]:
    """Patches whatever of the reference is importable; returns {target: "patched" | "skipped
    (...)"}.  Idempotent.  `force=True` ignores PIXELSPLAT_HIP=0."""
    report: dict[str, str] = {}
    if not (enabled() if force is None else force):
        return {"*": "disabled by PIXELSPLAT_HIP=0"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)      # makes the drop-in `diff_gaussian_rasterization` importable

    def rasterizer():
        import diff_gaussian_rasterization as drop_in
        assert drop_in.__file__.startswith(root), "another diff_gaussian_rasterization shadows the drop-in"

    def decoder():
        from .decoder import DecoderSplattingCUDA
        pkg = importlib.import_module("src.model.decoder")
        pkg.DECODERS["splatting_cuda"] = DecoderSplattingCUDA
        pkg.DecoderSplattingCUDA = DecoderSplattingCUDA

    def encoder_part(attr):
        def go():
            from . import encoder as ours
            mod = importlib.import_module("src.model.encoder.encoder_epipolar")
            setattr(mod, attr, getattr(ours, attr))
        return go

    def loss(wrapper_name, cls_name):
        def go():
            from . import loss as ours
            pkg = importlib.import_module("src.loss")
            wrapper = getattr(importlib.import_module(
                "src.loss." + {"LossMse": "loss_mse", "LossDepth": "loss_depth"}[cls_name]), wrapper_name)
            pkg.LOSSES[wrapper] = getattr(ours, cls_name)
            setattr(pkg, cls_name, getattr(ours, cls_name))
        return go

    _try(report, "diff_gaussian_rasterization", rasterizer)
    _try(report, "src.model.decoder.DECODERS[splatting_cuda]", decoder)
    for attr in ("EpipolarTransformer", "DepthPredictorMonocular", "GaussianAdapter"):
        _try(report, f"src.model.encoder.encoder_epipolar.{attr}", encoder_part(attr))
    _try(report, "src.loss.LOSSES[LossMseCfgWrapper]", loss("LossMseCfgWrapper", "LossMse"))
    _try(report, "src.loss.LOSSES[LossDepthCfgWrapper]", loss("LossDepthCfgWrapper", "LossDepth"))
    _APPLIED.update(report)
    return report


def applied() -> dict[str, str]:
    return dict(_APPLIED)


def main(argv: list[str]) -> int:
    """`python -m pixelsplat_amd.patch_reference <module> [args...]`: patch, then run <module> as
    __main__ with the remaining arguments (hydra reads sys.argv)."""
    import runpy

    if not argv:
        print(__doc__)
        return 2
    report = apply()
    for k, v in report.items():
        print(f"[pixelsplat_amd.patch_reference] {k}: {v}", file=sys.stderr)
    sys.argv = [argv[0], *argv[1:]]
    runpy.run_module(argv[0], run_name="__main__", alter_sys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
