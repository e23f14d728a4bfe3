"""Drop-in surface of the (A) module mirrors, checked on CPU: parameter / buffer names and
shapes identical to the reference's modules (so released checkpoints load), index tables
bit-identical, constructor signature.  The compute paths need the GPU (tests/test_epipolar_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_import as RI
from pixelsplat_amd.encoder import (EpipolarSampler, EpipolarTransformer, EpipolarTransformerCfg,
                                    ImageSelfAttentionCfg)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = dict(num_octaves=10, num_layers=2, num_heads=2, num_samples=4, d_dot=8, d_mlp=32, downscale=2)
SA = dict(patch_size=2, num_octaves=4, num_layers=1, num_heads=2, d_token=16, d_dot=8, d_mlp=32)


def _ours(v):
    return EpipolarTransformer(EpipolarTransformerCfg(self_attention=ImageSelfAttentionCfg(**SA),
                                                      **CFG), 16, num_context_views=v)


@pytest.mark.parametrize("name,v", [("transformer_v2.npz", 2), ("transformer_v3.npz", 3)])
def test_state_dict_of_the_reference_loads_strictly(name, v):
    z = np.load(os.path.join(GOLD, name))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    net = _ours(v)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert set(dict(net.named_parameters())) == set(sd)   # no extra parameters either


@pytest.mark.parametrize("v", [2, 3, 4])
def test_index_tables(v):
    s = EpipolarSampler(v, 8)
    idx = torch.tensor([[o for o in range(v) if o != i] for i in range(v)])
    assert torch.equal(s.index_v, idx)
    x = torch.arange(v * (v - 1)).reshape(1, v, v - 1)
    assert torch.equal(s.transpose(s.transpose(x)), x)          # an involution
    assert not any(k in s.state_dict() for k in ("index_v", "transpose_v", "transpose_ov"))
    if RI.available():
        m = RI.modules(v)
        r = m.sampler.EpipolarSampler(v, 8)
        for k in ("index_v", "transpose_v", "transpose_ov"):
            assert torch.equal(getattr(s, k), getattr(r, k))
        t = torch.randn(2, v, 5)
        assert torch.equal(s.collect(t), r.collect(t))


@pytest.mark.skipif(not RI.available(), reason="reference checkout not present (GPU box)")
def test_parameter_names_match_live_reference_at_paper_config():
    m = RI.modules(2)
    sa = dict(patch_size=4, num_octaves=10, num_layers=2, num_heads=4, d_token=128, d_dot=128,
              d_mlp=256)
    cf = dict(num_octaves=10, num_layers=2, num_heads=4, num_samples=32, d_dot=128, d_mlp=256,
              downscale=4)
    ref = m.transformer.EpipolarTransformer(m.transformer.EpipolarTransformerCfg(
        self_attention=m.self_attention.ImageSelfAttentionCfg(**sa), **cf), 128)
    ours = EpipolarTransformer(EpipolarTransformerCfg(
        self_attention=ImageSelfAttentionCfg(**sa), **cf), 128, num_context_views=2)
    a = {k: tuple(t.shape) for k, t in ref.state_dict().items()}
    b = {k: tuple(t.shape) for k, t in ours.state_dict().items()}
    assert a == b
    assert sum(p.numel() for p in ours.parameters()) == 6638848
