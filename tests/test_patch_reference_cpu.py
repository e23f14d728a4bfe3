"""pixelsplat_amd.patch_reference against the LIVE reference (build container only): the
registries and module namespaces `src/main.py` resolves at construction time
(/root/reference/src/model/encoder/__init__.py:8-20, src/model/decoder/__init__.py:5-13,
src/loss/__init__.py:6-16, encoder_epipolar.py:15-17) end up holding the HIP implementations
without a source edit, the swapped classes expose the reference's parameter names (released
checkpoints load), and PIXELSPLAT_HIP=0 switches everything off."""
import importlib
import os
import sys
import types
from types import SimpleNamespace

import pytest
import torch

from oracle import ref_import as RI

pytestmark = pytest.mark.skipif(not RI.available(), reason="needs /root/reference (build container)")


def _bare(name):
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(RI.REF, *name.split("."))]
    sys.modules[name] = m
    return m


@pytest.fixture()
def live_reference(monkeypatch):
    """The reference's real decoder / loss packages and encoder_epipolar module, with stand-ins
    only for third-party packages that are not installed here (torchvision, lpips, skimage) and
    for the dataset package (a type annotation)."""
    monkeypatch.delenv("PIXELSPLAT_HIP", raising=False)
    RI.setup(2)
    _bare("src.dataset").DatasetCfg = object
    _bare("src.dataset.shims")
    for name, attrs in (("lpips", ("LPIPS",)), ("torchvision", ()), ("torchvision.models", ("ResNet",)),
                        ("skimage", ()), ("skimage.metrics", ("structural_similarity",))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, object)
            sys.modules[name] = m
    for name in [n for n in sys.modules if n.startswith(("src.model.decoder", "src.loss",
                                                          "src.model.encoder.encoder_epipolar"))]:
        sys.modules.pop(name)
    sys.modules.pop("diff_gaussian_rasterization", None)
    yield
    for name in [n for n in sys.modules if n.startswith(("src.model.decoder", "src.loss"))]:
        sys.modules.pop(name)


def test_registries_are_swapped_without_source_edits(live_reference):
    from pixelsplat_amd import decoder as our_decoder
    from pixelsplat_amd import encoder as our_encoder
    from pixelsplat_amd import loss as our_loss
    from pixelsplat_amd import patch_reference as pr

    # the originals, before patching
    ref_enc = importlib.import_module("src.model.encoder.encoder_epipolar")
    originals = {k: getattr(ref_enc, k) for k in ("EpipolarTransformer", "DepthPredictorMonocular",
                                                   "GaussianAdapter")}
    report = pr.apply()
    assert all(v == "patched" for v in report.values()), report
    dec_pkg = importlib.import_module("src.model.decoder")
    loss_pkg = importlib.import_module("src.loss")
    assert dec_pkg.DECODERS["splatting_cuda"] is our_decoder.DecoderSplattingCUDA
    d = dec_pkg.get_decoder(SimpleNamespace(name="splatting_cuda"),
                            SimpleNamespace(background_color=[0.0, 0.0, 0.0]))
    assert isinstance(d, our_decoder.DecoderSplattingCUDA) and list(d.state_dict()) == []
    for k in originals:
        assert getattr(ref_enc, k) is getattr(our_encoder, k)
    mse_mod = importlib.import_module("src.loss.loss_mse")
    depth_mod = importlib.import_module("src.loss.loss_depth")
    losses = loss_pkg.get_losses([
        mse_mod.LossMseCfgWrapper(mse_mod.LossMseCfg(weight=2.0)),
        depth_mod.LossDepthCfgWrapper(depth_mod.LossDepthCfg(weight=0.25, sigma_image=None, use_second_derivative=False))])
    assert type(losses[0]) is our_loss.LossMse and losses[0].cfg.weight == 2.0 and losses[0].name == "mse"
    assert type(losses[1]) is our_loss.LossDepth and losses[1].name == "depth"
    assert sys.modules["diff_gaussian_rasterization"].__file__.startswith(
        os.path.dirname(os.path.dirname(os.path.abspath(pr.__file__))))

    # parameter / buffer names: what a released checkpoint's `encoder.*` keys must find
    m = RI.modules(2)
    cfg = m.transformer.EpipolarTransformerCfg(
        self_attention=m.self_attention.ImageSelfAttentionCfg(
            patch_size=4, num_octaves=10, num_layers=2, num_heads=4, d_token=128, d_dot=128, d_mlp=256),
        num_octaves=10, num_layers=2, num_heads=4, num_samples=32, d_dot=128, d_mlp=256, downscale=4)
    ref_t, our_t = originals["EpipolarTransformer"](cfg, 128), ref_enc.EpipolarTransformer(cfg, 128)
    assert {k: tuple(v.shape) for k, v in ref_t.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in our_t.state_dict().items()}
    assert sum(p.numel() for p in our_t.parameters()) == 6_638_848
    ref_d, our_d = originals["DepthPredictorMonocular"](128, 32, 1, False), \
        ref_enc.DepthPredictorMonocular(128, 32, 1, False)
    assert {k: tuple(v.shape) for k, v in ref_d.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in our_d.state_dict().items()}
    acfg = importlib.import_module("src.model.encoder.common.gaussian_adapter").GaussianAdapterCfg(0.5, 15.0, 4)
    ref_a, our_a = originals["GaussianAdapter"](acfg), ref_enc.GaussianAdapter(acfg)
    assert (ref_a.d_in, ref_a.d_sh) == (our_a.d_in, our_a.d_sh)
    assert list(ref_a.state_dict()) == list(our_a.state_dict())


def test_switch_off(live_reference, monkeypatch):
    from pixelsplat_amd import patch_reference as pr
    monkeypatch.setenv("PIXELSPLAT_HIP", "0")
    assert not pr.enabled()
    assert pr.apply() == {"*": "disabled by PIXELSPLAT_HIP=0"}
    dec_pkg = importlib.import_module("src.model.decoder")
    assert dec_pkg.DECODERS["splatting_cuda"].__module__ == "src.model.decoder.decoder_splatting_cuda"


def _calls_on(path, attr_names):
    """Every call `self.<attr>(...)`, `self.<attr>.<method>(...)` or `<attr>.<method>(...)` in a
    reference source file -> [(attr, method | None, n_positional, [keyword names], lineno)]."""
    import ast

    found = []
    for node in ast.walk(ast.parse(open(path).read())):
        if not isinstance(node, ast.Call):
            continue
        f, method = node.func, None
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Attribute) and f.value.attr in attr_names:
            f, method = f.value, f.attr                      # self.decoder.forward(...)
        elif isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in attr_names:
            found.append((f.value.id, f.attr, len(node.args), [k.arg for k in node.keywords], node.lineno))
            continue                                         # loss_fn.forward(...)
        if isinstance(f, ast.Attribute) and f.attr in attr_names and isinstance(f.value, ast.Name) \
                and f.value.id == "self":
            found.append((f.attr, method, len(node.args), [k.arg for k in node.keywords], node.lineno))
    return found


def test_every_reference_call_site_binds_to_the_drop_in_signatures(live_reference):
    """The glue nobody can run end to end (the reference is not on the GPU box, the GPU is not here):
    every call the reference makes into a swapped module -- encoder_epipolar.py:130-172 into the
    transformer, the depth predictor and the adapter; model_wrapper.py into the decoder and the losses --
    must bind to the drop-in's signature, by position and by keyword, and the constructors must take
    what the reference passes (encoder_epipolar.py:66-78)."""
    import inspect

    from pixelsplat_amd import decoder as our_decoder
    from pixelsplat_amd import encoder as our_encoder
    from pixelsplat_amd import loss as our_loss

    ref = RI.REF
    enc_calls = _calls_on(os.path.join(ref, "src/model/encoder/encoder_epipolar.py"),
                          {"epipolar_transformer", "depth_predictor", "gaussian_adapter"})
    mw_calls = _calls_on(os.path.join(ref, "src/model/model_wrapper.py"), {"decoder", "loss_fn"})
    owners = {"epipolar_transformer": our_encoder.EpipolarTransformer,
              "depth_predictor": our_encoder.DepthPredictorMonocular,
              "gaussian_adapter": our_encoder.GaussianAdapter,
              "decoder": our_decoder.DecoderSplattingCUDA}
    checked = 0
    for attr, method, n_pos, kw, line in enc_calls + mw_calls:
        if method in ("d_in", "d_sh") or (method is None and attr == "gaussian_adapter"):
            continue
        targets = [owners[attr]] if attr in owners else [our_loss.LossMse, our_loss.LossDepth]
        for cls in targets:
            fn = getattr(cls, method or "forward")
            try:
                inspect.signature(fn).bind(object(), *([object()] * n_pos), **{k: object() for k in kw})
            except TypeError as err:
                raise AssertionError(f"{attr}.{method or '__call__'} at line {line} does not bind to "
                                     f"{cls.__module__}.{cls.__name__}: {err}") from None
            checked += 1
    assert checked >= 8, (enc_calls, mw_calls)

    # forward / constructor parameter names of the reference classes are a positional prefix of ours
    ref_enc = importlib.import_module("src.model.encoder.encoder_epipolar")
    ref_dec = importlib.import_module("src.model.decoder.decoder_splatting_cuda")
    pairs = [(ref_enc.EpipolarTransformer, our_encoder.EpipolarTransformer),
             (ref_enc.DepthPredictorMonocular, our_encoder.DepthPredictorMonocular),
             (ref_enc.GaussianAdapter, our_encoder.GaussianAdapter),
             (ref_dec.DecoderSplattingCUDA, our_decoder.DecoderSplattingCUDA),
             (importlib.import_module("src.loss.loss_mse").LossMse, our_loss.LossMse),
             (importlib.import_module("src.loss.loss_depth").LossDepth, our_loss.LossDepth)]
    for ref_cls, our_cls in pairs:
        for name in ("__init__", "forward"):
            want = list(inspect.signature(getattr(ref_cls, name)).parameters)
            got = list(inspect.signature(getattr(our_cls, name)).parameters)
            assert got[:len(want)] == want, f"{our_cls.__name__}.{name}: {got} vs reference {want}"
            extra = list(inspect.signature(getattr(our_cls, name)).parameters.values())[len(want):]
            assert all(p.default is not inspect.Parameter.empty or p.kind in
                       (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in extra), f"{our_cls.__name__}.{name}: {extra}"
