"""One CONNECTED training step of the hot path: post-backbone features -> EpipolarTransformer.forward (the
full module: downscale conv, HIP epipolar layers, the image-self-attention feed-forward blocks and the
upscale / refinement convolutions PyTorch hosts) -> EncoderEpipolarHead (depth sampler, `to_gaussians`, Gaussian
adapter: HIP) -> DecoderSplattingCUDA.forward (HIP rasterizer) -> LossMse -> one backward to the features and
every weight.

Mirrors the reference's `ModelWrapper.training_step` (src/model/model_wrapper.py:108-152) from the point
where `EncoderEpipolar.forward` has its backbone features (src/model/encoder/encoder_epipolar.py:125-214); the
backbone (DINO / ResNet through torch.hub) and the high-resolution skip (a convolution of the context image)
are outside the hot path ("oracle input is the post-backbone feature tensor").  bench.py's headline
times the two kernel paths (A) and (B) on their own synthetic inputs (SURVEY.md 8d defines the metric so);
this module is what shows that the halves compose: tests/test_connected_gpu.py checks it against the REAL
reference chained on the CPU (tests/golden/connected.npz), `bench.py --connected` times it.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import torch
from torch import Tensor, nn

from .decoder import DecoderOutput, DecoderSplattingCUDA, DecoderSplattingCUDACfg
from .encoder import (EncoderEpipolarHead, EncoderEpipolarHeadCfg, EpipolarTransformer,
                      EpipolarTransformerCfg, GaussianAdapterCfg, Gaussians, OpacityMappingCfg)
from .loss import LossMse, LossMseCfg, LossMseCfgWrapper


@dataclass
class StepOutput:
    loss: Tensor
    color: Tensor                 # [b, v_tgt, 3, h, w]
    gaussians: Gaussians
    features: Tensor              # the epipolar transformer's output [b, v, c, h, w]


class ConnectedStep(nn.Module):
    """encoder (from the backbone features on) + decoder + loss, as the reference's training_step chains them."""

    def __init__(self, transformer_cfg: EpipolarTransformerCfg, d_feature: int, num_context_views: int,
                 head_cfg: EncoderEpipolarHeadCfg | None = None, background=(0.0, 0.0, 0.0),
                 mse_weight: float = 1.0) -> None:
        super().__init__()
        self.epipolar_transformer = EpipolarTransformer(transformer_cfg, d_feature,
                                                        num_context_views=num_context_views)
        if head_cfg is None:      # config/model/encoder/epipolar.yaml defaults
            head_cfg = EncoderEpipolarHeadCfg(
                d_feature=d_feature, num_monocular_samples=32, num_surfaces=1, predict_opacity=False,
                gaussians_per_pixel=3, gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, 4),
                opacity_mapping=OpacityMappingCfg(0.0, 0.0, 1), use_transmittance=False)
        self.head = EncoderEpipolarHead(head_cfg)
        self.decoder = DecoderSplattingCUDA(DecoderSplattingCUDACfg("splatting_cuda"),
                                            SimpleNamespace(background_color=list(background)))
        self.loss = LossMse(LossMseCfgWrapper(LossMseCfg(weight=mse_weight)))

    def encode(self, features: Tensor, context: dict, global_step: int = 0,
               deterministic: bool = False) -> tuple[Gaussians, Tensor]:
        """encoder_epipolar.py:125-214 without the backbone and the skip."""
        feats, _sampling = self.epipolar_transformer(
            features, context["extrinsics"], context["intrinsics"], context["near"], context["far"])
        return self.head(feats, context, global_step, deterministic), feats

    def forward(self, features: Tensor, context: dict, target: dict, global_step: int = 0,
                deterministic: bool = False) -> StepOutput:
        """model_wrapper.py:108-135: gaussians = encoder(...); output = decoder.forward(...);
        loss = sum of the losses (here LossMse, the re10k default besides LPIPS)."""
        gaussians, feats = self.encode(features, context, global_step, deterministic)
        h, w = target["image"].shape[-2:]
        out: DecoderOutput = self.decoder.forward(
            gaussians, target["extrinsics"], target["intrinsics"], target["near"], target["far"], (h, w),
            depth_mode=None)
        loss = self.loss.forward(out, {"target": target}, gaussians, global_step)
        return StepOutput(loss, out.color, gaussians, feats)
