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


def test_fold_attention_weights_algebra():
    """Host logic of path (A): the folded matrices reproduce, head by head, the unfused chain
    q = W_q x, q~ = W_k^T q, u = W_d^T q~, e = E q~ on the input side and
    y = sum_h W_o,h W_v,h (fbar + W_d pbar + E^T abar + b_d) + b_o on the output side
    (attention.py:54-70 with kv = features + Linear(PE) + view embedding)."""
    from pixelsplat_amd.epipolar import fold_attention_weights

    torch.manual_seed(0)
    H, dh, c, d, P, ov, d_out = 3, 5, 8, 6, 4, 2, 7
    wq, wkv = torch.randn(H * dh, d), torch.randn(2 * H * dh, c)
    wo, bo = torch.randn(d_out, H * dh), torch.randn(d_out)
    dw, db, ve = torch.randn(c, P), torch.randn(c), torch.randn(ov, c)
    for emb in (ve, None):
        w_in, w_o_t, bias = fold_attention_weights(w_q=wq, w_kv=wkv, w_out=wo, b_out=bo, heads=H,
                                                   depth_w=dw, depth_b=db, view_emb=emb)
        lh = w_in.shape[0] // H
        assert lh % 4 == 0 and w_o_t.shape == (H * lh, d_out)
        x = torch.randn(d)
        wk = wkv[:H * dh].reshape(H, dh, c)
        wv = wkv[H * dh:].reshape(H, dh, c)
        fb, pb, ab = torch.randn(H, c), torch.randn(H, P), torch.randn(H, ov)
        fused = torch.zeros(H * lh)
        ref = bo.clone()
        for h in range(H):
            qt = wk[h].T @ (wq.reshape(H, dh, d)[h] @ x)
            row = w_in[h * lh:(h + 1) * lh] @ x
            assert torch.allclose(row[:c], qt, atol=1e-4)
            assert torch.allclose(row[c:c + P], dw.T @ qt, atol=1e-4)
            if emb is not None:
                assert torch.allclose(row[c + P:c + P + ov], ve @ qt, atol=1e-4)
            fused[h * lh:h * lh + c] = fb[h]
            fused[h * lh + c:h * lh + c + P] = pb[h]
            ctxv = fb[h] + dw @ pb[h] + db
            if emb is not None:
                fused[h * lh + c + P:h * lh + c + P + ov] = ab[h]
                ctxv = ctxv + ve.T @ ab[h]
            ref += wo.reshape(d_out, H, dh)[:, h] @ (wv[h] @ ctxv)
        assert torch.allclose(fused @ w_o_t + bias, ref, atol=1e-3)


def test_sampling_features_are_gathered_on_first_access():
    """EpipolarSampling keeps the reference's constructor and attribute names; `features` left out by the
    fused path is produced once, on first read, by the deferred gather."""
    from pixelsplat_amd.encoder import EpipolarSampling

    z = torch.zeros(1)
    calls = []

    def gather():
        calls.append(1)
        return torch.full((2, 3), 7.0)

    s = EpipolarSampling(features=None, valid=z, xy_ray=z, xy_sample=z, xy_sample_near=z, xy_sample_far=z,
                         origins=z, directions=z, lazy_features=gather)
    assert not s.features_materialized and "lazy" in repr(s) and calls == []
    f = s.features
    assert f.shape == (2, 3) and s.features is f and calls == [1] and s.features_materialized
    eager = EpipolarSampling(torch.ones(4), z, z, z, z, z, z, z, lazy_features=gather)
    assert eager.features.shape == (4,) and calls == [1]          # a given tensor wins, no gather
    none = EpipolarSampling(None, z, z, z, z, z, z, z)
    assert none.features is None
    none.features = torch.ones(2)
    assert none.features.shape == (2,)
    assert set(EpipolarSampling.FIELDS) == {"features", "valid", "xy_ray", "xy_sample", "xy_sample_near",
                                            "xy_sample_far", "origins", "directions"}


def test_overlap_mask_is_unpacked_on_first_read():
    """`EpipolarGeometry.overlaps` (= `EpipolarSampling.valid`, epipolar_sampler.py:66-75) is bit 0 of the
    flag byte the geometry kernel writes; the kernels of the hot path read the flags themselves, so the bool
    tensor is only made when somebody asks for it -- once."""
    from pixelsplat_amd.epipolar import EpipolarGeometry

    z = torch.zeros(1)
    flags = torch.tensor([[0, 1, 2, 3, 7, 6]], dtype=torch.uint8)
    geo = EpipolarGeometry(z, z, z, z, z, z, flags, z, z, z)
    assert geo._overlaps is None
    m = geo.overlaps
    assert m.dtype == torch.bool and m.tolist() == [[False, True, False, True, True, False]]
    assert geo.overlaps is m


@pytest.mark.parametrize("c,octaves,v,has_e,lh,pad", [
    (128, 10, 2, False, 148, 0),      # BASELINE configs[1]: 592 = 4 x 148, nothing to zero
    (128, 10, 3, True, 152, 2),       # configs[3]: two floats behind the view term of each head
    (32, 9, 2, False, 52, 2), (16, 10, 4, True, 40, 1), (64, 10, 2, True, 88, 3)])
def test_head_stride_padding_is_handed_to_the_kernels(c, octaves, v, has_e, lh, pad):
    """The attention kernels zero the padding behind each head's last block themselves
    (PsEpipolarDesc.tail_pad_in / tail_pad_out); the descriptor the Python side builds must say how much."""
    from pixelsplat_amd.epipolar import _FusedEpipolarAttention as F

    d, width = F._desc((1, v, 4, 4, 8, c, 4, octaves), has_e)
    assert width == lh == d.hs_in == d.hs_out and lh % 4 == 0
    assert d.tail_pad_in == d.tail_pad_out == pad
    assert d.ld_q == d.ld_f == 4 * lh
