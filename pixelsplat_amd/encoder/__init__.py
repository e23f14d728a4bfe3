"""Drop-in mirrors of the reference's epipolar encoder modules (path A), same class names,
constructor arguments, parameter / buffer names (released checkpoints load unchanged) and
return types; the sampler geometry, feature gather and cross-attention run on the HIP
kernels of libpixelsplat_hip.so."""
from .depth_predictor import DepthPredictorMonocular, sample_depths  # noqa: F401
from .encoder_epipolar import (EncoderEpipolarHead, EncoderEpipolarHeadCfg,  # noqa: F401
                               OpacityMappingCfg)
from .epipolar_sampler import EpipolarSampler, EpipolarSampling  # noqa: F401
from .epipolar_transformer import (EpipolarTransformer, EpipolarTransformerCfg,  # noqa: F401
                                   ImageSelfAttentionCfg)
from .gaussian_adapter import GaussianAdapter, GaussianAdapterCfg, Gaussians  # noqa: F401
from .transformer import Attention, FeedForward, PreNorm, Transformer  # noqa: F401
